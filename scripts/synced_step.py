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

# where a synchronised step goes: the call (host work + launch), the wait, and the same with 1000 grains (kernel ~3 us:
# what remains is the launch-to-completion latency of the runtime)
def split(view_, label):
    vr = ctypes.byref(view_)
    tc = tw = 0.
    steps = 3000
    for _ in range(steps):
        a = time.perf_counter()
        lib.rebxamd_wh_steps(ctx._h, vr, arr, len(ops), dt, 1, 0, sp)
        b = time.perf_counter()
        hipc.hipStreamSynchronize(sp)
        c = time.perf_counter()
        tc += b - a; tw += c - b
    print("%-14s call %.2f us  wait %.2f us  total %.2f us" % (label, tc / steps * 1e6, tw / steps * 1e6, (tc + tw) / steps * 1e6))


split(view, "1e6 grains")
small = {k: v[:1001] for k, v in host.items()}
dps = DeviceParticles.from_host(small, device="cuda:0", n_head=1, tail_massless=True)
split(dps.view(), "1000 grains")
t0 = time.perf_counter()
for _ in range(3000):
    lib.rebxamd_wh_steps(ctx._h, ctypes.byref(dps.view()) if False else vref, arr, len(ops), dt, 1, 0, sp)
hipc.hipStreamSynchronize(sp)
print("queued single-step calls, 1e6 grains: %.2f us per call" % ((time.perf_counter() - t0) / 3000 * 1e6))
vs = dps.view(); vsr = ctypes.byref(vs)
t0 = time.perf_counter()
for _ in range(3000):
    lib.rebxamd_wh_steps(ctx._h, vsr, arr, len(ops), dt, 1, 0, sp)
hipc.hipStreamSynchronize(sp)
print("queued single-step calls, 1000 grains: %.2f us per call (host cost of a call when the device keeps up)" % ((time.perf_counter() - t0) / 3000 * 1e6))


def split_open(view_, label):
    vr = ctypes.byref(view_)
    lib.rebxamd_wh_run_begin(ctx._h, vr, arr, len(ops), dt, 0, sp)
    tc = tw = 0.
    steps = 3000
    for _ in range(steps):
        a = time.perf_counter()
        lib.rebxamd_wh_run_steps(ctx._h, 1, sp)
        b = time.perf_counter()
        hipc.hipStreamSynchronize(sp)
        c = time.perf_counter()
        tc += b - a; tw += c - b
    lib.rebxamd_wh_run_end(ctx._h, sp)
    hipc.hipStreamSynchronize(sp)
    print("open run, %-14s call %.2f us  wait %.2f us  total %.2f us" % (label, tc / steps * 1e6, tw / steps * 1e6, (tc + tw) / steps * 1e6))


split_open(view, "1e6 grains")
split_open(vs, "1000 grains")


def split_open_query(view_, label):
    """the same, waiting by polling hipStreamQuery instead of hipStreamSynchronize"""
    vr = ctypes.byref(view_)
    lib.rebxamd_wh_run_begin(ctx._h, vr, arr, len(ops), dt, 0, sp)
    steps = 3000
    t0 = time.perf_counter()
    for _ in range(steps):
        lib.rebxamd_wh_run_steps(ctx._h, 1, sp)
        while hipc.hipStreamQuery(sp) != 0:
            pass
    tot = time.perf_counter() - t0
    lib.rebxamd_wh_run_end(ctx._h, sp)
    hipc.hipStreamSynchronize(sp)
    print("open run, %-14s polling hipStreamQuery: total %.2f us" % (label, tot / steps * 1e6))


split_open_query(view, "1e6 grains")
split_open_query(vs, "1000 grains")
