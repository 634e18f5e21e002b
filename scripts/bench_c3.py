"""Times the all-massive stack of BASELINE config 3 on one GPU (development aid):
tides_precession + modify_orbits_forces + gr_potential, every grain with mass 1e-15, so the
pushes on the star are parallel reductions; PARTICLE coordinates (streaming path) and the
reference's default Jacobi coordinates (prefix-sum path).
usage: python scripts/bench_c3.py [n] [reps]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from reboundx_amd import synthetic  # noqa: E402
from reboundx_amd.device import Context, DeviceParticles  # noqa: E402
from reboundx_amd.program import build_ops  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
C = synthetic.C_DEFAULT
ctx = Context(0)
t0 = time.time()
cols, pp, _ = synthetic.disk(n, 0, dust_mass=1e-15)
print("generated %d particles in %.1f s" % (n, time.time() - t0), flush=True)
pp["primary"] = np.zeros(n, dtype=np.uint8); pp["primary"][0] = 1
host = dict(cols)
for k in ("R_tides", "k1", "tau_a", "tau_e", "tau_inc"):
    host[k] = pp[k]
dp = DeviceParticles.from_host(host, n_head=1, tail_massless=False)
del cols, host
view = dp.view()
for label, coords in (("PARTICLE", 2), ("JACOBI", 0)):
    stack = [("tides_precession", {}), ("modify_orbits_forces", {"coordinates": coords}), ("gr_potential", {"c": C})]
    ops = build_ops(stack, {k: v[:1] for k, v in pp.items()})
    for mode in (0,):
        for _ in range(2):
            ctx.apply(view, ops, mode)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.apply(view, ops, mode)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("C3 %-8s n=%d  %9.1f us/step  %.3e evals/s  %6.2f TB/s algorithmic at 144 B (%.1f%% of 8 TB/s)" % (
            label, n, dt * 1e6, n / dt, 144 * n / dt / 1e12, 144 * n / dt / 1e12 / 0.08), flush=True)
