"""Latency of one drop-in call per force, operator and diagnostic on the reference's three-body test system
(reboundx/test/test_machine_independent.py) and on a 1000-particle disk: what a REBOUND user pays per call.
usage: python scripts/hook_latency_each.py"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from reboundx_amd import synthetic  # noqa: E402
from reboundx_amd.extras import REBX_INTEGRATORS, Extras, Simulation  # noqa: E402

FORCES = {
    "radiation_forces": {"c": 1e4}, "gravitational_harmonics": {}, "gr": {"c": 1e4}, "gr_potential": {"c": 1e4}, "gr_full": {"c": 1e4},
    "tides_precession": {}, "central_force": {}, "modify_orbits_forces": {},
}


def build(n):
    cols = synthetic.twoplanets() if n == 3 else synthetic.disk(n, 2)[0]
    sim = Simulation()
    for i in range(n):
        sim.add(m=cols["m"][i], x=cols["x"][i], y=cols["y"][i], z=cols["z"][i], vx=cols["vx"][i], vy=cols["vy"][i], vz=cols["vz"][i])
    sim.integrator = 1
    sim.dt = 0.01
    rebx = Extras(sim)
    ps = sim.particles
    ps[0].params["J2"], ps[0].params["J4"], ps[0].params["R_eq"] = 1e-3, 1e-3, 1e-3
    ps[0].params["R_tides"], ps[0].params["k1"] = 1e-3, 0.4
    ps[0].params["Acentral"], ps[0].params["gammacentral"] = 1e-3, -1.0
    ps[0].params["radiation_source"] = 1
    for i in range(1, min(n, 3)):
        ps[i].params["beta"] = 0.3
        ps[i].params["tau_a"] = -1e4
        ps[i].params["tau_e"] = -1e3
        ps[i].params["tau_mass"] = -1e6
    return sim, rebx


def timed(call, reps):
    for _ in range(20):
        call()
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    return (time.perf_counter() - t0) / reps * 1e6


for n in (3, 1000):
    reps = 1000
    for name, ep in FORCES.items():
        sim, rebx = build(n)
        f = rebx.load_force(name)
        for k, v in ep.items():
            f.params[k] = v
        rebx.add_force(f)
        print("n=%5d  force     %-26s %7.1f us per call" % (n, name, timed(sim.call_additional_forces, reps if name != "gr_full" else 200)), flush=True)
        sim.free_particle_params()
    for name in ("modify_mass", "modify_orbits_direct"):
        sim, rebx = build(n)
        op = rebx.load_operator(name)
        print("n=%5d  operator  %-26s %7.1f us per call" % (n, name, timed(lambda: op.step(sim, 1e-6), reps)), flush=True)
        sim.free_particle_params()
    for integ in ("euler", "rk4", "implicit_midpoint"):
        sim, rebx = build(n)
        f = rebx.load_force("modify_orbits_forces")
        op = rebx.load_operator("integrate_force")
        op.params["force"] = f
        op.params["integrator"] = REBX_INTEGRATORS[integ]
        print("n=%5d  operator  %-26s %7.1f us per call" % (n, "integrate_force/" + integ, timed(lambda: op.step(sim, 1e-6), reps)), flush=True)
        sim.free_particle_params()
    sim, rebx = build(n)
    gr = rebx.load_force("gr"); gr.params["c"] = 1e4
    grp = rebx.load_force("gr_potential"); grp.params["c"] = 1e4
    grf = rebx.load_force("gr_full"); grf.params["c"] = 1e4
    for name, call in (("gravitational_harmonics", rebx.gravitational_harmonics_potential), ("central_force", rebx.central_force_potential),
                       ("tides_precession", rebx.tides_precession_potential), ("gr_potential", lambda: rebx.gr_potential_potential(grp)),
                       ("gr_hamiltonian", lambda: rebx.gr_hamiltonian(gr)), ("gr_full_hamiltonian", lambda: rebx.gr_full_hamiltonian(grf))):
        print("n=%5d  potential %-26s %7.1f us per call" % (n, name, timed(call, 300)), flush=True)
    sim.free_particle_params()
