"""Latency of one host-synchronised Wisdom-Holman step (star + 1e6 grains, five effects) under the runtime's two ways of
waiting for a stream: blocking (default) and spinning (hipDeviceScheduleSpin).  usage: python scripts/synced_step.py [spin]"""
import ctypes
import os
import sys
import time

spin = len(sys.argv) > 1 and sys.argv[1] == "spin"
if spin:
    hip = ctypes.CDLL("libamdhip64.so")
    rc = hip.hipSetDeviceFlags(ctypes.c_uint(1))          # hipDeviceScheduleSpin, before anything creates the context
    print("hipSetDeviceFlags(spin) ->", rc)

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from reboundx_amd import synthetic  # noqa: E402
from reboundx_amd._lib import Op, lib  # noqa: E402
from reboundx_amd.device import Context, DeviceParticles  # noqa: E402
from reboundx_amd.program import build_ops  # noqa: E402

C = synthetic.C_DEFAULT
stack = [("radiation_forces", {"c": C}), ("gravitational_harmonics", {}), ("gr_potential", {"c": C}), ("tides_precession", {}), ("modify_orbits_forces", {})]
n = 1_000_000
cols, pp, _ = synthetic.disk(n + 1, 0)
host = dict(cols)
for k in ("beta", "R_tides", "k1", "tau_a", "tau_e", "tau_inc"):
    host[k] = pp[k]
ctx = Context(0)
dp = DeviceParticles.from_host(host, device="cuda:0", n_head=1, tail_massless=True)
ops = build_ops(stack, {k: v[:1] for k, v in pp.items()})
view = dp.view()
dt = 2 * np.pi / 100.0
arr = (Op * len(ops))(*ops)
stream = torch.cuda.current_stream()
vref, sp = ctypes.byref(view), ctypes.c_void_p(stream.cuda_stream)
hipc = ctypes.CDLL("libamdhip64.so")
for label, wait in (("torch stream.synchronize", stream.synchronize), ("hipStreamSynchronize", lambda: hipc.hipStreamSynchronize(sp))):
    for _ in range(200):
        lib.rebxamd_wh_steps(ctx._h, vref, arr, len(ops), dt, 1, 0, sp); wait()
    t0 = time.perf_counter()
    steps = 5000
    for _ in range(steps):
        lib.rebxamd_wh_steps(ctx._h, vref, arr, len(ops), dt, 1, 0, sp)
        wait()
    print("%-28s %s: %.2f us per synchronised step" % (label, "spin" if spin else "default", (time.perf_counter() - t0) / steps * 1e6))
