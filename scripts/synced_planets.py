"""Latency of a host-synchronised Wisdom-Holman step behind a head of several massive bodies (development aid): star +
`planets` planets + 10^6 grains under radiation forces; closed steps (rebxamd_wh_steps(..., 1) + a stream synchronisation)
and a run left open between the host's waits (rebxamd_wh_run_steps(1) + a stream synchronisation).
usage: python scripts/synced_planets.py"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reboundx_amd import synthetic  # noqa: E402
from reboundx_amd._lib import Op, lib  # noqa: E402
from reboundx_amd.device import Context, DeviceParticles  # noqa: E402
from reboundx_amd.program import build_ops  # noqa: E402

C = synthetic.C_DEFAULT
ctx = Context(0)
n = 1_000_000
for planets in (0, 1, 4):
    cols, pp, _ = synthetic.disk(n + 1 + planets, planets)
    host = dict(cols)
    host["beta"] = pp["beta"]
    dp = DeviceParticles.from_host(host, device="cuda:0", n_head=1 + planets, tail_massless=True)
    ops = build_ops([("radiation_forces", {"c": C})], {k: v[:1 + planets] for k, v in pp.items()})
    view = dp.view()
    arr = (Op * len(ops))(*ops)
    vref, stream = ctypes.byref(view), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    dt = 2 * np.pi / 100
    for _ in range(20):
        lib.rebxamd_wh_steps(ctx._h, vref, arr, len(ops), dt, 1, 0, stream)
    torch.cuda.synchronize()
    steps = 2000
    t0 = time.perf_counter()
    for _ in range(steps):
        rc = lib.rebxamd_wh_steps(ctx._h, vref, arr, len(ops), dt, 1, 0, stream)
        torch.cuda.current_stream().synchronize()
    t = (time.perf_counter() - t0) / steps
    print("planets %d: %.1f us per synchronised step (rc %d)" % (planets, t * 1e6, rc))
    rc = lib.rebxamd_wh_run_begin(ctx._h, vref, arr, len(ops), dt, 0, stream)
    assert rc == 0, rc
    for _ in range(20):
        lib.rebxamd_wh_run_steps(ctx._h, 1, stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        rc = lib.rebxamd_wh_run_steps(ctx._h, 1, stream)
        torch.cuda.current_stream().synchronize()
    t = (time.perf_counter() - t0) / steps
    lib.rebxamd_wh_run_end(ctx._h, stream)
    torch.cuda.synchronize()
    print("planets %d: %.1f us per synchronised step of an open run (rc %d)" % (planets, t * 1e6, rc))
