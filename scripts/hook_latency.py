"""Latency of the drop-in call (sim->additional_forces, host array-of-structs in and out) on small systems.
usage: python scripts/hook_latency.py [grfull] [n ...]      (grfull: gr_full alone instead of gr + harmonics + radiation; REBXAMD_HOOK_IN_PLACE=0: one copy per direction instead of in-place evaluation)"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from reboundx_amd import synthetic  # noqa: E402
from reboundx_amd.extras import Extras, Particle, Simulation  # noqa: E402

args = list(sys.argv[1:])
stack = ("gr_full",) if "grfull" in args else ("gr", "gravitational_harmonics", "radiation_forces")
args = [a for a in args if a != "grfull"]
sizes = [int(float(a)) for a in args] or [100, 1000, 4000, 10000]
for n in sizes:
    cols, pp, _ = synthetic.disk(n, 1, dust_mass=1e-9) if stack == ("gr_full",) else synthetic.disk(n, 1)
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
    for name in stack:
        f = rebx.load_force(name)
        if name != "gravitational_harmonics":
            f.params["c"] = synthetic.C_DEFAULT
        rebx.add_force(f)
    arr[0].params["J2"] = 1e-3
    arr[0].params["J4"] = 1e-3
    arr[0].params["R_eq"] = 1e-3
    arr[0].params["radiation_source"] = 1
    rebx.set_param_column("beta", pp["beta"])
    for _ in range(50):
        sim.call_additional_forces()
    reps = 2000 if stack != ("gr_full",) else 300
    t0 = time.perf_counter()
    for _ in range(reps):
        sim.call_additional_forces()
    dt = (time.perf_counter() - t0) / reps
    print("n=%6d  %.1f us per call  (in place: %s)" % (n, dt * 1e6, os.environ.get("REBXAMD_HOOK_IN_PLACE", "1") != "0"), flush=True)
    del rebx, sim
