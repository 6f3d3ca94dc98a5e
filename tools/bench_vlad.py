"""NetVLAD-FC head timing at the reference's size (64 x 512, 48x48 positions -> 512-D)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gloc3d_amd import capi
rng = np.random.default_rng(0)
K, C, HW, OUT = 64, 512, 48 * 48, 512
m = capi.NetVladFC((rng.standard_normal((K, C)) * 0.2).astype(np.float32), rng.random((K, C)).astype(np.float32),
                   (rng.standard_normal((K * C, OUT)) / 22).astype(np.float32))
m.set_profile(True)
for n in (1, 8, 64):
    x = torch.relu(torch.randn(n, C, HW, device="cuda")); out = torch.empty(n, OUT, device="cuda")
    for _ in range(3): m.forward_device(x.data_ptr(), n, HW, out.data_ptr())
    torch.cuda.synchronize(); t = time.time()
    R = 20
    for _ in range(R): m.forward_device(x.data_ptr(), n, HW, out.data_ptr())
    capi.check(capi.lib().gloc_vlad_set_stream(m._h, None)); dt = (time.time() - t) / R
    print(f"n={n}: {dt*1e6:.0f} us/call = {dt/n*1e6:.1f} us/descriptor")
