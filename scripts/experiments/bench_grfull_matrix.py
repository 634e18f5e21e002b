# NOTE: needs scripts/experiments/grfull_matrix_cores.patch applied (the variant it measures was not kept).
"""gr_full passes with the pairs' inner products on the matrix cores against the same passes on the vector pipe alone:
time per pass and largest relative difference of the results (development aid).  usage: python scripts/bench_grfull_matrix.py [n]"""
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
pot = z()
an = [dp.cols[k].clone() for k in ("ax", "ay", "az")]
ao = [0.01 * a for a in an]
flag = torch.zeros(1, dtype=torch.int32, device="cuda")
c = 1e4
pairs = n * (n - 1.0)
ctx.grfull_pot(view, c, 0, n, pot)
res = {}
for matrix in (False, True):
    ctx.grfull_matrix_cores(matrix)
    ac, out = [z(), z(), z()], [z(), z(), z()]
    for name, flops, fn in (("const", 60, lambda: ctx.grfull_const(view, c, 0, n, pot, ac)),
                            ("sweep", 36, lambda: ctx.grfull_sweep(view, c, 0, n, an, ao, ac, out, flag))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("%-7s %-6s %8.3f ms  %6.2f TFLOP/s (%4.1f%% of 78.6)" % ("matrix" if matrix else "vector", name, dt * 1e3, pairs * flops / dt / 1e12, pairs * flops / dt / 1e12 / 0.786), flush=True)
    res[matrix] = (torch.stack(ac).cpu().numpy(), torch.stack(out).cpu().numpy())
    dq = DeviceParticles.from_host(cl, n_head=n, tail_massless=False)
    for k, s in (("ax", "x"), ("ay", "y"), ("az", "z")):
        dq.cols[k].copy_(-dq.cols[s])
    ctx.grfull(dq.view(), c, 0, 0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        sw = ctx.grfull(dq.view(), c, 0, 0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("%-7s full evaluation: %d sweeps, %.2f ms, %.2f TFLOP/s" % ("matrix" if matrix else "vector", sw, dt * 1e3, pairs * (70 + 36 * sw) / dt / 1e12), flush=True)
for k, name in enumerate(("const", "sweep")):
    a, b = res[False][k], res[True][k]
    big = np.abs(a).max(axis=0)
    print(name, "max |matrix - vector| / largest component of the row: %.3e" % (np.abs(a - b).max(axis=0) / big).max())
