"""Development aid (GPU box): BASELINE configuration 5 at its real length -- star + 1e6 massless grains, five effects,
1e4 Wisdom-Holman steps in batches -- and how far a sample of grains ends from the checker's run of (star + sample).
python tests/tools/c5_length.py [n] [steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import oracle
from reboundx_amd import synthetic
from reboundx_amd.device import Context, DeviceParticles
from reboundx_amd.program import build_ops
from test_gpu_wh import C5_STACK

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
dt = 2 * np.pi / 100.0
ctx = Context(0)
cols, pp, _ = synthetic.disk(n + 1, 0)
sample = np.concatenate([[0], 1 + np.random.default_rng(5).choice(n, size=256, replace=False)])
sample.sort()
for mode in (0, 1):
    host = dict(cols)
    for k in ("beta", "R_tides", "k1", "tau_a", "tau_e", "tau_inc"):
        host[k] = pp[k]
    dp = DeviceParticles.from_host(host, device="cuda:0", n_head=1, tail_massless=True)
    ops = build_ops(C5_STACK, {k: v[:1] for k, v in pp.items()})
    t0 = time.perf_counter()
    ctx.wh_steps(dp.view(), ops, dt, steps, mode)
    torch.cuda.synchronize()
    print("mode", mode, "device %.2f s" % (time.perf_counter() - t0), flush=True)
    got = np.array([dp.cols[k][torch.from_numpy(sample).cuda()].cpu().numpy() for k in ("x", "y", "z", "vx", "vy", "vz")])
    sub = {k: v[sample].copy() for k, v in cols.items()}
    spp = {k: v[sample].copy() for k, v in pp.items()}
    ref = oracle.Sim(sub)
    t0 = time.perf_counter()
    for chunk in (100, 900, 9000) if steps == 10_000 else (steps,):
        oracle.whfast_steps(ref, dt, chunk, C5_STACK, spp)
    print("  checker %.2f s" % (time.perf_counter() - t0), flush=True)
    want = np.array([ref.cols[k] for k in ("x", "y", "z", "vx", "vy", "vz")])
    r = np.sqrt((want[:3] ** 2).sum(axis=0))
    dev = np.abs(got - want).max(axis=0)
    print("  max |device - checker| %.3e (state scale %.3g); relative to the grain's distance %.3e; unbound or far grains: %d" % (
        dev.max(), np.abs(want).max(), (dev / np.maximum(r, 1.)).max(), int((r > 5).sum())), flush=True)
    order = np.argsort(-dev)[:5]
    print("  worst grains", [(int(sample[i]), float(dev[i]), float(r[i])) for i in order])
