"""reboundx/test/test_simulationarchive.py re-stated on the device-resident path: gr (c = 100) on star + planet (m = 1e-4,
a = 1, e = 0.2), dt = P/100, as a force in the kick of a Wisdom-Holman step and as an integrate_force operator before, after
and around the step (euler, rk2, rk4, implicit_midpoint).  As in the reference's test: the pericentre has turned by more
than 0.1 rad at t = 1000, the gr Hamiltonian is conserved to 1e-4, and a run continued from an intermediate snapshot -- the
state copied off the device, the force's parameters saved to a .rebx file and read back -- lands on the same bits as the run
that never stopped.  (REBOUND's SimulationArchive holds the particle state there; here it is a copy of the columns.)"""
import os

import numpy as np
import pytest

from reboundx_amd import synthetic
from reboundx_amd.device import DeviceParticles
from reboundx_amd.extras import Extras, Simulation
from reboundx_amd.program import build_ops

pytestmark = pytest.mark.gpu
C = 1e2
STATE = ("x", "y", "z", "vx", "vy", "vz")


@pytest.fixture(scope="module")
def ctx():
    from reboundx_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def star_and_planet():
    m = np.array([1.0, 1e-4])
    x, y, z, vx, vy, vz = synthetic.kepler_to_cartesian(m.sum(), 1.0, 0.2, 0.0, 0.0, 0.0, 0.0)
    cols = {"x": np.array([0.0, x]), "y": np.array([0.0, y]), "z": np.array([0.0, z]),
            "vx": np.array([0.0, vx]), "vy": np.array([0.0, vy]), "vz": np.array([0.0, vz]), "m": m}
    for k in STATE:
        cols[k] = cols[k] - (m * cols[k]).sum() / m.sum()
    for k in ("ax", "ay", "az"):
        cols[k] = np.zeros(2)
    period = 2 * np.pi / np.sqrt(m.sum())
    return cols, 1e-2 * period


def pomega(s, m):
    """longitude of pericentre of body 1 about body 0 (planar orbit: angle of the eccentricity vector)"""
    r, v = s[:3, 1] - s[:3, 0], s[3:, 1] - s[3:, 0]
    mu = m.sum()
    e = ((v @ v - mu / np.sqrt(r @ r)) * r - (r @ v) * v) / mu
    return float(np.arctan2(e[1], e[0]))


def host_state(dp):
    return np.array([dp.cols[k].cpu().numpy() for k in STATE])


def device_particles(cols, s=None):
    host = dict(cols)
    if s is not None:
        for k, row in zip(STATE, s):
            host[k] = row.copy()
    return DeviceParticles.from_host(host, device="cuda:0", n_head=2, tail_massless=True)


def ops_through_a_rebx_file(tmp_path, c):
    """The force's parameter as the reference's test carries it across the snapshot: set through the plugin API, saved with
    rebx_output_binary, read back with rebx_create_extras_from_binary."""
    sim = Simulation()
    sim.add(m=1.0)
    rebx = Extras(sim)
    gr = rebx.load_force("gr")
    gr.params["c"] = c
    rebx.add_force(gr)
    path = os.path.join(str(tmp_path), "test.rebx")
    rebx.save(path)
    sim2 = Simulation()
    sim2.add(m=1.0)
    back = Extras(sim2, path)
    c_back = back.get_force("gr").params["c"]
    assert c_back == c
    return build_ops([("gr", {"c": c_back})], {})


def advance(ctx, dp, ops, dt, nsteps, mode, timing, scheme):
    view = dp.view()
    if timing == "force":
        ctx.wh_steps(view, ops, dt, nsteps, mode)
        return
    for _ in range(nsteps):
        if timing == "pre":
            ctx.integrate_force(view, ops, scheme, dt, mode)
        elif timing == "both":
            ctx.integrate_force(view, ops, scheme, 0.5 * dt, mode)
        ctx.wh_steps(view, [], dt, 1, mode)
        if timing == "post":
            ctx.integrate_force(view, ops, scheme, dt, mode)
        elif timing == "both":
            ctx.integrate_force(view, ops, scheme, 0.5 * dt, mode)


CASES = [("force", None, 1), ("force", None, 0)] + [(t, s, 0) for t in ("pre", "post", "both") for s in ("euler", "rk2", "rk4", "implicit_midpoint")]


@pytest.mark.parametrize("timing,scheme,mode", CASES, ids=["%s-%s-%s" % (t, s or "kick", "exact" if m else "fast") for t, s, m in CASES])
def test_gr_precession_energy_and_restart_from_a_snapshot(ctx, tmp_path, timing, scheme, mode):
    cols, dt = star_and_planet()
    m = cols["m"]
    ops = build_ops([("gr", {"c": C})], {})
    dp = device_particles(cols)
    assert abs(pomega(host_state(dp), m)) < 1e-12
    e0 = ctx.gr_hamiltonian(dp.view(), C)
    n1 = int(round(1000.0 / dt))
    # the operator cases make three launches' worth of calls per step from Python: a third of the run is enough for the restart
    n2 = 2 * n1 if timing == "force" else n1 // 3
    advance(ctx, dp, ops, dt, n1, mode, timing, scheme)
    snapshot = host_state(dp)
    assert np.isfinite(snapshot).all()
    assert pomega(snapshot, m) > 0.1                                         # the effect turns the pericentre
    assert abs((ctx.gr_hamiltonian(dp.view(), C) - e0) / e0) < 1e-4          # ... and conserves its Hamiltonian
    advance(ctx, dp, ops, dt, n2, mode, timing, scheme)
    end = host_state(dp)
    again = device_particles(cols, snapshot)
    advance(ctx, again, ops_through_a_rebx_file(tmp_path, C), dt, n2, mode, timing, scheme)
    assert np.array_equal(host_state(again), end)
    assert np.abs(end - snapshot).max() > 1e-3
