"""Development aid (GPU box): many drop-in calls in a row on small systems, evaluated in the pinned host columns and
through copies side by side; every call must give the same bits (a result read before it has landed in host memory
would show here).   python tests/tools/inplace_stress.py [calls]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from reboundx_amd import synthetic
from reboundx_amd.extras import Extras, Particle, Simulation

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
bad = 0
for n in (3, 40, 1000, 9000):
    cols, pp, _ = synthetic.disk(n, 2)
    sims = []
    for in_place in (True, False):
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
        rebx.set_in_place(in_place)
        for name in ("gr", "gravitational_harmonics", "radiation_forces"):
            f = rebx.load_force(name)
            if name != "gravitational_harmonics":
                f.params["c"] = synthetic.C_DEFAULT
            rebx.add_force(f)
        arr[0].params["J2"] = 1e-3
        arr[0].params["J4"] = 1e-3
        arr[0].params["R_eq"] = 1e-3
        arr[0].params["radiation_source"] = 1
        rebx.set_param_column("beta", pp["beta"])
        sims.append((sim, rebx, view))
    rng = np.random.default_rng(n)
    for it in range(calls if n <= 1000 else calls // 5):
        i = int(rng.integers(1, n))
        dx = 1e-6 * rng.standard_normal()
        for sim, _, view in sims:
            view[i, 0] += dx
            view[:, 6:9] = 0.
            sim.call_additional_forces()
        a, b = sims[0][2][:, 6:9], sims[1][2][:, 6:9]
        if not np.array_equal(a.view(np.uint64), b.view(np.uint64)) or not np.abs(a[1:]).max() > 0:
            bad += 1
            print("n", n, "call", it, "differs on", int((a != b).any(axis=1).sum()), "bodies", flush=True)
            if bad > 10:
                sys.exit(1)
    print("n", n, "done, calls that differ so far:", bad, flush=True)
sys.exit(1 if bad else 0)
