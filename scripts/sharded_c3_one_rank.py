"""Config 3 (Jacobi and PARTICLE coordinates) through the particle-sharded driver on ONE rank, next to rebxamd_apply on the same
columns (development aid): what the driver's extra passes and small launches cost before any collective.
usage: python scripts/sharded_c3_one_rank.py [n]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reboundx_amd import synthetic  # noqa: E402
from reboundx_amd.device import Context, DeviceParticles  # noqa: E402
from reboundx_amd.dist import apply_forces_sharded  # noqa: E402
from reboundx_amd.program import build_ops  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
C = synthetic.C_DEFAULT
ctx = Context(0)
cols = synthetic.disk_device(1 + n, "cuda:0", dust_mass=1e-15, seed=3)
keep = ("x", "y", "z", "vx", "vy", "vz", "m", "ax", "ay", "az", "R_tides", "k1", "tau_a", "tau_e", "tau_inc")
dp = DeviceParticles({k: cols[k] for k in keep}, n_head=1, tail_massless=False)
sp = synthetic.star_params()
for label, coords in (("particle", 2), ("jacobi", 0)):
    ops = build_ops([("tides_precession", {}), ("modify_orbits_forces", {"coordinates": coords}), ("gr_potential", {"c": C})], sp)
    for name, run in (("rebxamd_apply", lambda: ctx.apply(dp.view(), ops, 0)), ("sharded driver", lambda: apply_forces_sharded(ctx, dp, ops, 0))):
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            run()
        torch.cuda.synchronize()
        print("%-9s %-15s %8.1f us per evaluation" % (label, name, (time.perf_counter() - t0) / 10 * 1e6), flush=True)
