"""Developer probe: the scan index as the search sees it (sortedness, permutation), and the time of indexing a batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gloc3d_amd import capi, synth
w = synth.make_world(1001)
A = np.ascontiguousarray(synth.lidar_scan(w, None, seed=1001)[:, :3])
st = capi.ScanStore()
for n in (100, 2048, 2049, 5000, 40000, len(A)):
    c = np.ascontiguousarray(A[:n])
    i = st.add(c)
    d = st.debug_index(i)
    u = np.unique(d["perm"])
    print(n, "unique", len(u), "min", d["perm"].min(), "max", d["perm"].max(), "keys sorted", bool((np.diff(d["keys"].astype(np.int64)) >= 0).all()),
          "first", d["perm"][:6], "dl ok", bool((st.download(i) == c).all()), flush=True)
scans = [np.ascontiguousarray(A[::1] + np.float32(0.01 * k)) for k in range(25)]
for rep in range(3):
    t0 = time.time(); ids = st.add_batch(scans); t1 = time.time()
    print(f"add_batch of 25 x {len(A)}: {(t1 - t0) * 1e3:.2f} ms", flush=True)
    t0 = time.time(); one = [st.add(s) for s in scans[:5]]; t1 = time.time()
    print(f"add x 5 one by one: {(t1 - t0) * 1e3 / 5:.3f} ms each", flush=True)
    if rep == 0:
        t0 = time.time(); st.build_target_index_batch(ids); t1 = time.time()
        print(f"kd batch of 25: {(t1 - t0) * 1e3:.2f} ms = {(t1 - t0) * 1e3 / 25:.3f} ms per scan", flush=True)
        t0 = time.time(); st.build_target_index(one[0]); t1 = time.time()
        print(f"kd of one scan: {(t1 - t0) * 1e3:.2f} ms", flush=True)
    for i in ids + one:
        st.release(i)
