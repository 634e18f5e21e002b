"""Times the streaming kernels for the stacked-force configurations (development aid).
usage: python scripts/bench_stream.py [n] [reps]"""
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
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
C = synthetic.C_DEFAULT
ctx = Context(0)
if os.environ.get("FIXED") == "0":          # walk the op list even for stacks with an instance of their own
    ctx.fixed_stacks(False)

CONFIGS = {
    "C2 rad+J2J4+gr (112 B)": (0.0, 112, [("radiation_forces", {"c": C}), ("gravitational_harmonics", {}), ("gr", {"c": C})]),
    "C5 rad+J2J4+grpot+tides+mof (152 B)": (0.0, 152, [("radiation_forces", {"c": C}), ("gravitational_harmonics", {}), ("gr_potential", {"c": C}),
                                                       ("tides_precession", {}), ("modify_orbits_forces", {"coordinates": 2})]),
    "C3 tides+mof+grpot massive (144 B)": (1e-15, 144, [("tides_precession", {}), ("modify_orbits_forces", {"coordinates": 2}), ("gr_potential", {"c": C})]),
    "rad only (104 B)": (0.0, 104, [("radiation_forces", {"c": C})]),
    "gr only (104 B)": (0.0, 104, [("gr", {"c": C})]),
    "harmonics only (80 B)": (0.0, 80, [("gravitational_harmonics", {})]),
}
only = os.environ.get("ONLY")
for name, (dust_mass, bpe, stack) in CONFIGS.items():
    if only and only not in name:
        continue
    cols, pp, _ = synthetic.disk(n, 0, dust_mass=dust_mass)
    pp["primary"] = np.zeros(n, dtype=np.uint8); pp["primary"][0] = 1
    host = dict(cols)
    for k in ("beta", "R_tides", "k1", "tau_a", "tau_e", "tau_inc"):
        host[k] = pp[k]
    dp = DeviceParticles.from_host(host)
    ops = build_ops(stack, {k: v[:1] for k, v in pp.items()})
    view = dp.view()
    for mode in (0, 1):
        for _ in range(3):
            ctx.apply(view, ops, mode)
        torch.cuda.synchronize()
        ctx.timing_enable(True)
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.apply(view, ops, mode)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps
        ms, cnt = ctx.timing_read()
        ctx.timing_enable(False)
        us = ms / cnt * 1e3
        print("%-40s %-5s kernel %8.1f us  wall/step %8.1f us  %7.1f GB/s algorithmic (%.1f%% of 8 TB/s)  %.3e evals/s" % (
            name, "fast" if mode == 0 else "exact", us, wall * 1e6, bpe * n / us / 1e3, bpe * n / us / 1e3 / 80.0, n / wall), flush=True)
    del dp
