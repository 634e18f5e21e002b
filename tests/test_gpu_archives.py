"""The drop-in path against the reference's stored trajectories.

For every force the reference's test suite keeps a parameter file (<force>.rebx) and a
trajectory archive (<force>.sa) of its three-body system under WHFast.  Here the parameter
file is loaded with this library's reader, the force it names is evaluated ON THE GPU through
the sim->additional_forces hook at every one of the 15 917 steps to the archive's second
snapshot, and the state must land on the archived one to roundoff.  The integrator around the
hook is the checker's Wisdom-Holman restatement, which reproduces the force-free archive by
itself (tests/test_oracle.py)."""
import ctypes
import json
import os
import warnings

import numpy as np
import pytest

import oracle
from reboundx_amd.extras import Extras, Simulation

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FORCES = ["central_force", "gr", "gr_full", "gr_potential", "gravitational_harmonics", "modify_orbits_forces", "radiation_forces",
          "tides_precession"]


@pytest.fixture(scope="module")
def archives():
    return json.load(open(os.path.join(GOLDEN, "reference_archives.json")))


def rows(entry):
    return np.array([[float.fromhex(v) for v in row] for row in entry["particles"]])


@pytest.mark.parametrize("mode", ["exact", "fast"])
@pytest.mark.parametrize("name", FORCES)
def test_replay_with_the_force_evaluated_on_the_device(archives, name, mode):
    start, ref = rows(archives["forces"][name][0]), rows(archives["forces"][name][1])
    dt = float.fromhex(archives["dt"])
    steps = round(float.fromhex(archives["forces"][name][1]["t"]) / dt)
    sim = Simulation(integrator="whfast")
    for p in start:
        sim.add(m=p[6], x=p[0], y=p[1], z=p[2], vx=p[3], vy=p[4], vz=p[5])
    sim.gravity_ignore_terms = 1                   # what WHFast tells the forces (gr_full needs it)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")            # the fixture was written by an older version
        rebx = Extras(sim, os.path.join(GOLDEN, "rebx", name + ".rebx"))
    rebx.set_mode(mode)
    assert sim.additional_forces
    ps = sim.particles
    n = 3

    def hook(ptr, _user):
        s = ptr.contents
        cols = [np.ctypeslib.as_array(ctypes.cast(getattr(s, k), ctypes.POINTER(ctypes.c_double)), shape=(n,))
                for k in ("x", "y", "z", "vx", "vy", "vz", "ax", "ay", "az")]
        for i in range(n):
            p = ps[i]
            p.x, p.y, p.z, p.vx, p.vy, p.vz, p.ax, p.ay, p.az = (float(c[i]) for c in cols)
        sim.additional_forces(ctypes.byref(sim))
        for i in range(n):
            cols[6][i], cols[7][i], cols[8][i] = ps[i].ax, ps[i].ay, ps[i].az

    cols = {k: start[:, j].copy() for j, k in enumerate(("x", "y", "z", "vx", "vy", "vz", "m"))}
    for k in ("ax", "ay", "az"):
        cols[k] = np.zeros(n)
    state = oracle.Sim(cols)
    cb = oracle._FORCE_FN(hook)
    rc = oracle.lib.orc_whfast_steps(ctypes.byref(state.s), dt, steps, cb, oracle._STEP_FN(0), oracle._STEP_FN(0), None)
    assert rc == 0
    sim.process_messages()
    got = np.array([state.cols[k] for k in ("x", "y", "z", "vx", "vy", "vz")]).T
    scale = max(1.0, np.abs(ref[:, :6]).max())
    assert np.abs(got - ref[:, :6]).max() <= 3e-10 * scale
    sim.free_particle_params()


@pytest.mark.parametrize("name", ["modify_mass", "modify_orbits_direct"])
def test_replay_with_the_operator_run_on_the_device(archives, name):
    """Same for the two operators: their fixture puts half a step before and half a step after
    every timestep; REBOUND would call sim->pre/post_timestep_modifications, here the checker's
    integrator does.  Masses (modify_mass) are part of the compared state."""
    start, ref = rows(archives["forces"][name][0]), rows(archives["forces"][name][1])
    dt = float.fromhex(archives["dt"])
    steps = round(float.fromhex(archives["forces"][name][1]["t"]) / dt)
    sim = Simulation(integrator="whfast")
    for p in start:
        sim.add(m=p[6], x=p[0], y=p[1], z=p[2], vx=p[3], vy=p[4], vz=p[5])
    sim.dt = dt
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rebx = Extras(sim, os.path.join(GOLDEN, "rebx", name + ".rebx"))
    rebx.set_mode("exact")
    ps = sim.particles
    n = 3
    fields = ("x", "y", "z", "vx", "vy", "vz", "m")

    def run(hook_fn):
        def step(ptr, _h, _user):
            s = ptr.contents
            cols = [np.ctypeslib.as_array(ctypes.cast(getattr(s, k), ctypes.POINTER(ctypes.c_double)), shape=(n,)) for k in fields]
            for i in range(n):
                for k, c in zip(fields, cols):
                    setattr(ps[i], k, float(c[i]))
            hook_fn(ctypes.byref(sim))
            for i in range(n):
                for k, c in zip(fields, cols):
                    c[i] = getattr(ps[i], k)
        return oracle._STEP_FN(step)

    cols = {k: start[:, j].copy() for j, k in enumerate(fields)}
    for k in ("ax", "ay", "az"):
        cols[k] = np.zeros(n)
    state = oracle.Sim(cols)
    pre, post = run(sim.pre_timestep_modifications), run(sim.post_timestep_modifications)
    rc = oracle.lib.orc_whfast_steps(ctypes.byref(state.s), dt, steps, oracle._FORCE_FN(0), pre, post, None)
    assert rc == 0
    sim.process_messages()
    got = np.array([state.cols[k] for k in fields]).T
    assert np.abs(got - ref).max() <= 2e-9
    assert np.abs(ref[:, :6] - start[:, :6]).max() > 0.1
    sim.free_particle_params()


@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_replay_of_the_integrate_force_archive_with_the_operator_run_on_the_device(archives, mode):
    """test.sa / test.rebx (reboundx/test/test_simulationarchive.py:56-84): star + planet under WHFast from
    t = 1000 to t = 2000 (15 916 steps of dt and a shortened last one) with the integrate_force operator over
    gr, c = 100, ahead of every step; no `integrator` parameter in the file: the default scheme, Euler
    (src/integrate_force.c:40-44).  The operator comes out of the reference's own parameter file through
    rebx_init_extras_from_binary and runs on the device through sim->pre_timestep_modifications."""
    entry = archives["integrate_force"]
    start, ref = rows(entry["snapshots"][0]), rows(entry["snapshots"][1])
    dt, t0, t1 = float.fromhex(entry["dt"]), float.fromhex(entry["snapshots"][0]["t"]), float.fromhex(entry["snapshots"][1]["t"])
    sim = Simulation(integrator="whfast")
    for p in start:
        sim.add(m=p[6], x=p[0], y=p[1], z=p[2], vx=p[3], vy=p[4], vz=p[5])
    sim.dt = dt
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rebx = Extras(sim, os.path.join(GOLDEN, "rebx", "test.rebx"))
    rebx.set_mode(mode)
    assert sim.pre_timestep_modifications and not sim.post_timestep_modifications and not sim.additional_forces
    ps = sim.particles
    n = 2
    fields = ("x", "y", "z", "vx", "vy", "vz", "m")

    def pre(ptr, _h, _user):
        s = ptr.contents
        cols = [np.ctypeslib.as_array(ctypes.cast(getattr(s, k), ctypes.POINTER(ctypes.c_double)), shape=(n,)) for k in fields]
        for i in range(n):
            for k, c in zip(fields, cols):
                setattr(ps[i], k, float(c[i]))
        sim.pre_timestep_modifications(ctypes.byref(sim))
        for i in range(n):
            for k, c in zip(fields, cols):
                c[i] = getattr(ps[i], k)

    cols = {k: start[:, j].copy() for j, k in enumerate(fields)}
    for k in ("ax", "ay", "az"):
        cols[k] = np.zeros(n)
    state = oracle.Sim(cols)
    t, whole = t0, 0
    while t + dt < t1:
        t = (t + dt / 2) + dt / 2
        whole += 1
    cb = oracle._STEP_FN(pre)
    assert oracle.lib.orc_whfast_steps(ctypes.byref(state.s), dt, whole, oracle._FORCE_FN(0), cb, oracle._STEP_FN(0), None) == 0
    sim.dt = t1 - t                                   # the operator acts for sim->dt * dt_fraction (src/core.c:882-896)
    assert oracle.lib.orc_whfast_steps(ctypes.byref(state.s), t1 - t, 1, oracle._FORCE_FN(0), cb, oracle._STEP_FN(0), None) == 0
    sim.process_messages()
    got = np.array([state.cols[k] for k in fields]).T
    assert whole == 15916
    assert np.abs(got - ref).max() <= 3e-10
    assert np.abs(ref[:, :6] - start[:, :6]).max() > 0.1
    sim.free_particle_params()
