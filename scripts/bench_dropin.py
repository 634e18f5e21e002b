"""Times the drop-in path (host array-of-structs in and out through sim->additional_forces):
pack, host-to-device, kernels, device-to-host + unpack, as reported by rebxamd_last_timings.
usage: python scripts/bench_dropin.py [n] [gr | j2 | beta-lists | beta-column]
  gr           gr + gravitational_harmonics: no per-particle parameters beyond the star's
  j2           gravitational_harmonics alone: a position-only batch, which uploads no velocities
  beta-lists   radiation_forces with beta set on every particle by rebx_set_param_double (one list per particle:
               every call reads n values again through the cached pointers)
  beta-column  the same beta given whole by rebxamd_set_param_column (no lists)"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from reboundx_amd import synthetic  # noqa: E402
from reboundx_amd.extras import Extras, Particle, Simulation, lib  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
C = synthetic.C_DEFAULT
cols, pp, _ = synthetic.disk(n, 0)
sim = Simulation()
arr = (Particle * n)()
view = np.frombuffer(arr, dtype=np.float64).reshape(n, 16)
for j, k in enumerate(("x", "y", "z", "vx", "vy", "vz", "ax", "ay", "az", "m")):
    view[:, j] = cols[k]
sim._store = arr
sim._particles = ctypes.cast(arr, ctypes.POINTER(Particle))
sim.N = n
arr[0]._sim = ctypes.pointer(sim)
rebx = Extras(sim)
what = sys.argv[2] if len(sys.argv) > 2 else "gr"
if what in ("gr", "j2"):
    for name in ("gr", "gravitational_harmonics") if what == "gr" else ("gravitational_harmonics",):          # executed: harmonics, then gr
        f = rebx.load_force(name)
        if name == "gr":
            f.params["c"] = C
        rebx.add_force(f)
    arr[0].params["J2"] = 1e-3
    arr[0].params["J4"] = 1e-3
    arr[0].params["R_eq"] = 1e-3
else:
    f = rebx.load_force("radiation_forces")
    f.params["c"] = C
    rebx.add_force(f)
    t0 = time.perf_counter()
    if what == "beta-column":
        rebx.set_param_column("beta", pp["beta"])
    else:
        ap0, stride = ctypes.addressof(arr) + Particle.ap.offset, ctypes.sizeof(Particle)
        set_double, me, key = lib.rebx_set_param_double, ctypes.byref(rebx), b"beta"
        beta = pp["beta"]
        for i in range(1, n):
            set_double(me, ctypes.cast(ap0 + i * stride, ctypes.POINTER(ctypes.c_void_p)), key, float(beta[i]))
    print("%s: parameters set in %.2f s" % (what, time.perf_counter() - t0), flush=True)
for k in range(4):
    t0 = time.perf_counter()
    sim.call_additional_forces()
    wall = time.perf_counter() - t0
    t = rebx.last_timings()
    # pack and upload overlap block by block, as do download and unpack: "h2d" is what remains of the upload after
    # the last block is packed
    print("call %d  n=%d  wall %8.2f ms | pack %7.2f  h2d wait %7.2f  kernels %6.3f  d2h+unpack %7.2f ms | %.3e evals/s end to end (48 B in once the masses are there, 24 B out per particle; 80 B in with the exact arithmetic)" % (
        k, n, wall * 1e3, t["pack_ms"], t["h2d_ms"], t["kernel_ms"], t["d2h_unpack_ms"], n / wall), flush=True)
