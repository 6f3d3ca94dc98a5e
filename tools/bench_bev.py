"""BEV projection stage benchmark ("next" row N1): batches of synthetic scans, device resident.
Not the bench.py line: numbers for DESIGN.md and rocprof."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gloc3d_amd import capi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--scans", type=int, default=64); ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--format", type=int, default=1); ap.add_argument("--cpu", type=int, default=1)
a = ap.parse_args()
w = synth.make_world(7)
base = [np.ascontiguousarray(synth.lidar_scan(w, synth.se3(3.0 * i, (2.0 * i, 1.0 * i, 0)), 100 + i)[:, :3]) for i in range(8)]
clouds = [base[i % 8] for i in range(a.scans)]
off = np.concatenate([[0], np.cumsum([c.shape[0] for c in clouds])]).astype(np.uint64)
d_xyz = torch.from_numpy(np.concatenate(clouds)).cuda()
p = capi.default_bev_params(format=a.format)
shape = (a.scans, 768, 768, 3) if a.format == 0 else (a.scans, 3, 768, 768)
d_out = torch.empty(shape, dtype=torch.uint8 if a.format == 0 else torch.float32, device="cuda")
proj = capi.BevProjector()
for B in sorted({1, a.scans}):
    o = off[:B + 1]
    for _ in range(5): proj.project_batch_device(d_xyz.data_ptr(), o, 3, d_out.data_ptr(), p, want_info=False)
    proj.synchronize(); t = time.time()
    for _ in range(a.reps): proj.project_batch_device(d_xyz.data_ptr(), o, 3, d_out.data_ptr(), p, want_info=False)
    proj.synchronize(); dt = (time.time() - t) / a.reps
    npts = int(o[-1]); S = 2 * (500 + 2) + 1
    alg = 24.0 * npts + B * (1.0 * S * S + d_out[0].numel() * d_out.element_size())
    print(f"BEV batch {B}: {dt*1e6:.1f} us/batch -> {B/dt:.0f} scans/s, {npts/dt/1e9:.2f} Gpoint/s, "
          f"algorithmic {alg/dt/1e12:.2f} TB/s (points in twice + flags cleared + image out)")
proj.set_profile(True)
for _ in range(20): proj.project_batch_device(d_xyz.data_ptr(), off, 3, d_out.data_ptr(), p, want_info=False)
for k in ("bev_clear", "bev_mark", "bev_flag", "bev_image"):
    ms, n = proj.profile(k); print(f"  {k}: {ms/n*1e3:.1f} us/launch")
if a.cpu:
    import oracle
    t = time.time(); n = 0
    while time.time() - t < 3.0:
        oracle.bev_crop_pad(oracle.bev_project(base[n % 8])[0]); n += 1
    print(f"CPU long form (oracle, 1 core): {(time.time()-t)/n*1e3:.1f} ms/scan")
