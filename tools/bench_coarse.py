"""Coarse (x, y, yaw) match stage benchmark: one query grid against 20 database grids (device resident)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gloc3d_amd import capi, synth
w = synth.make_world(1001)
cm = capi.CoarseMatcher()
t0 = time.time()
db = [cm.add_scan(synth.lidar_scan(w, synth.se3(7.0 * i, (1.5 * i, -0.8 * i, 0.0)), seed=10 + i, n_az=1000)) for i in range(20)]
q = cm.add_scan(synth.lidar_scan(w, synth.se3(33.0, (4.0, 3.0, 0.0)), seed=99, n_az=1000))
print("cells per grid:", [len(cm.cells(g)) for g in db[:4]], len(cm.cells(q)))
qs = synth.lidar_scan(w, synth.se3(33.0, (4.0, 3.0, 0.0)), seed=99)
t = time.time()
for _ in range(20): g = cm.add_scan(qs); cm.release(g)
print(f"add_scan (BEV projection + grid): {(time.time() - t) / 20 * 1e3:.2f} ms per 124k-point scan (host pointer in, synchronous)")
for n in (1, 20):
    cm.match(q, db[:n])
    t = time.time()
    for _ in range(50): xy, r, ok = cm.match(q, db[:n])
    dt = (time.time() - t) / 50
    print(f"match 1 query x {n} places (360 yaw x 129 lags x 2 axes + 13 x 81 2-D checks each): {dt * 1e3:.3f} ms -> {dt / n * 1e6:.0f} us per pair")
print("first result", xy[0], r[0], ok[0])
