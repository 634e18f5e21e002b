"""Times gr_full on one GPU (development aid).  usage: python scripts/bench_grfull.py [n]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from reboundx_amd import synthetic  # noqa: E402
from reboundx_amd.device import Context, DeviceParticles  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 65536
ctx = Context(0)
cl = synthetic.cluster(n)
dp = DeviceParticles.from_host(cl, n_head=n, tail_massless=False)
for k, s in (("ax", "x"), ("ay", "y"), ("az", "z")):
    dp.cols[k].copy_(-dp.cols[s])
view = dp.view()
z = lambda: torch.zeros(n, dtype=torch.float64, device="cuda")
pot, ac, out = z(), [z(), z(), z()], [z(), z(), z()]
an = [dp.cols[k].clone() for k in ("ax", "ay", "az")]
ao = [z(), z(), z()]
flag = torch.zeros(1, dtype=torch.int32, device="cuda")
c = 1e4
pairs = n * (n - 1.0)
for name, flops, fn in (("pot", 10, lambda: ctx.grfull_pot(view, c, 0, n, pot)),
                        ("const", 60, lambda: ctx.grfull_const(view, c, 0, n, pot, ac)),
                        ("sweep", 36, lambda: ctx.grfull_sweep(view, c, 0, n, an, ao, ac, out, flag))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("%-6s %8.3f ms  %7.2f Gpairs/s  %6.2f TFLOP/s (%4.1f%% of 78.6)" % (name, dt * 1e3, pairs / dt / 1e9, pairs * flops / dt / 1e12, pairs * flops / dt / 1e12 / 0.786))
t0 = time.perf_counter()
sw = ctx.grfull(view, c, 0, 0)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("full evaluation: %d sweeps, %.2f ms, %.3e pair-interactions/s" % (sw, dt * 1e3, pairs * (2 + sw) / dt))
