"""The "integrate a force across a timestep" operator and the kick/drift steppers on the GPU
(rebxamd_integrate_force / rebxamd_kick / rebxamd_drift through the C ABI) against the CPU
checker's restatement of src/integrate_force.c and src/integrator_*.c.

Bars: reference-order arithmetic (MODE_EXACT) on a star + planets + massless-dust layout is
bitwise equal to the checker for the explicit schemes (everything is elementwise); the default
arithmetic and layouts with parallel reductions meet 1e-12 relative to the velocity change of
the particle (the quantity the operator computes; the velocity itself is O(1) and would hide
errors)."""
import numpy as np
import pytest

import oracle
from gpu_common import TOL, ulp_diff
from reboundx_amd import synthetic
from reboundx_amd.device import DeviceParticles
from reboundx_amd.program import build_ops

pytestmark = pytest.mark.gpu
C = synthetic.C_DEFAULT
EXACT, FAST = 1, 0
C5_STACK = [("radiation_forces", {"c": C}), ("gravitational_harmonics", {}), ("gr_potential", {"c": C}),
            ("tides_precession", {}), ("modify_orbits_forces", {})]


@pytest.fixture(scope="module")
def ctx():
    from reboundx_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def device_state(cols, pp, **kw):
    host = dict(cols)
    for k in ("beta", "R_tides", "k1", "tau_a", "tau_e", "tau_inc"):
        if pp.get(k) is not None:
            host[k] = pp[k]
    return DeviceParticles.from_host(host, device="cuda:0", **kw)


def vel(dp):
    return np.array([dp.cols[k].cpu().numpy() for k in ("vx", "vy", "vz")])


def run_pair(ctx, cols, pp, forces, integrator, dt, mode, **kw):
    ref = oracle.Sim(cols)
    n_ref = oracle.integrate_force(ref, integrator, dt, forces, pp)
    dp = device_state(cols, pp, **kw)
    n_dev = ctx.integrate_force(dp.view(), build_ops(forces, pp), integrator, dt, mode)
    v0 = np.array([cols[k] for k in ("vx", "vy", "vz")])
    vr = np.array([ref.cols[k] for k in ("vx", "vy", "vz")])
    return v0, vr, vel(dp), np.array(ref.acc), np.array(dp.acc_host()), n_ref, n_dev


def dv_err(v0, vr, vd):
    dv = vr - v0
    scale = np.abs(dv).max(axis=0)
    scale = np.where(scale == 0., 1., scale)
    # the update is formed as v + dt*a: it cannot carry more digits than eps*|v| allows
    floor = 4 * np.finfo(float).eps * np.abs(vr).max(axis=0)
    return (np.abs(vd - vr) / np.maximum(scale, floor / TOL)[None, :]).max()


@pytest.mark.parametrize("mode", [EXACT, FAST])
@pytest.mark.parametrize("integrator", ["euler", "rk2", "rk4", "implicit_midpoint"])
def test_five_effect_stack_star_planets_dust(ctx, integrator, mode):
    n = 6001
    cols, pp, _ = synthetic.disk(n, 2)
    v0, vr, vd, ar, ad, n_ref, n_dev = run_pair(ctx, cols, pp, C5_STACK, integrator, 0.01, mode)
    assert n_ref == n_dev
    if mode == EXACT and integrator != "implicit_midpoint":
        assert ulp_diff(vd, vr).max() == 0
        assert ulp_diff(ad, ar).max() == 0            # what is left in the acceleration columns (k1)
    assert dv_err(v0, vr, vd) <= TOL
    if integrator == "implicit_midpoint":
        assert 1 <= n_dev < 10 and not ad.any() and not ar.any()


C2_STACK = [("radiation_forces", {"c": C}), ("gravitational_harmonics", {}), ("gr", {"c": C})]
MIXED_STACK = [("modify_orbits_forces", {"coordinates": 1}), ("radiation_forces", {"c": C}), ("gr", {"c": C}), ("tides_precession", {})]


@pytest.mark.parametrize("mode", [EXACT, FAST])
@pytest.mark.parametrize("integrator", ["euler", "rk2", "rk4", "implicit_midpoint"])
@pytest.mark.parametrize("forces", [C5_STACK, C2_STACK, MIXED_STACK], ids=["five-effects", "rad+harmonics+gr", "walked-op-list"])
@pytest.mark.parametrize("n,planets", [(2, 0), (5, 1), (4099, 2), (6002, 3)])
def test_update_carried_by_one_pass_equals_separate_passes(ctx, n, planets, forces, integrator, mode):
    """With massless bodies behind a short massive prefix, one pass over them evaluates every
    stage of the scheme in registers (the prefix runs the separate passes, one head table per
    stage).  Switched off, every stage is a force pass plus an elementwise pass.  Same
    expressions: bit-equal velocities and left-over accelerations in reference-order arithmetic.
    The implicit scheme is carried round by round (same number of rounds either way)."""
    cols, pp, _ = synthetic.disk(n, planets)
    ops = build_ops(forces, pp)
    out, rounds = [], []
    for carried in (True, False):
        dp = device_state(cols, pp)
        try:
            ctx.carried_steps(carried)
            rounds.append(ctx.integrate_force(dp.view(), ops, integrator, 0.01, mode))
        finally:
            ctx.carried_steps(True)
        out.append((vel(dp), np.array(dp.acc_host())))
    (v1, a1), (v2, a2) = out
    assert rounds[0] == rounds[1]
    if integrator == "implicit_midpoint":
        assert 1 <= rounds[0] < 10 and not a1.any() and not a2.any()
    assert np.isfinite(v1).all() and np.isfinite(a1).all()
    if mode == EXACT:
        assert np.array_equal(v1, v2) and np.array_equal(a1, a2)
    else:
        v0 = np.array([cols[k] for k in ("vx", "vy", "vz")])
        assert dv_err(v0, v2, v1) <= TOL
        big = np.abs(a2).max(axis=0)
        assert (np.abs(a1 - a2) / np.where(big == 0., 1., big)[None, :]).max() <= TOL


@pytest.mark.parametrize("mode", [EXACT, FAST])
@pytest.mark.parametrize("integrator", ["euler", "rk2", "rk4", "implicit_midpoint"])
@pytest.mark.parametrize("n", [2, 3, 17, 64, 65])
def test_small_systems_updated_in_one_launch_equal_the_separate_passes(ctx, n, integrator, mode):
    """Up to 64 bodies, every one of them massive: all stages (or rounds) of the update run in one launch of one
    wavefront; switched off (with the carried passes), every stage is a force pass plus an elementwise pass.  Same
    functions in the same order: bit-equal velocities, left-over accelerations and round counts in reference-order
    arithmetic, and the checker's velocities within the bar.  65 bodies take the separate passes either way."""
    cols, pp, _ = synthetic.disk(n, min(3, n - 1), dust_mass=1e-7)
    pp = dict(pp)
    pp["primary"] = np.zeros(n, dtype=np.uint8)
    pp["primary"][0] = 1
    forces = [("modify_orbits_forces", {"coordinates": 2}), ("gr", {"c": C}), ("modify_orbits_forces", {}), ("radiation_forces", {"c": C}), ("tides_precession", {})]
    ops = build_ops(forces, pp)
    ref = oracle.Sim(cols)
    n_ref = oracle.integrate_force(ref, integrator, 0.01, forces, pp)
    out, rounds = [], []
    for one_launch in (True, False):
        dp = device_state(cols, pp, n_head=n, tail_massless=False)
        try:
            ctx.carried_steps(one_launch)
            rounds.append(ctx.integrate_force(dp.view(), ops, integrator, 0.01, mode))
        finally:
            ctx.carried_steps(True)
        out.append((vel(dp), np.array(dp.acc_host())))
    (v1, a1), (v2, a2) = out
    assert rounds[0] == rounds[1] == n_ref
    v0 = np.array([cols[k] for k in ("vx", "vy", "vz")])
    vr = np.array([ref.cols[k] for k in ("vx", "vy", "vz")])
    assert dv_err(v0, vr, v1) <= TOL
    if mode == EXACT:
        assert np.array_equal(v1, v2) and np.array_equal(a1, a2)
        if integrator != "implicit_midpoint":
            assert ulp_diff(v1, vr).max() == 0
    else:
        assert dv_err(v0, v2, v1) <= TOL


@pytest.mark.parametrize("integrator", ["euler", "rk4", "implicit_midpoint"])
@pytest.mark.parametrize("odd", [("vx",), ("vy", "vz"), ("x", "vx", "ax")])
def test_columns_at_odd_offsets_take_the_eight_byte_accesses(ctx, integrator, odd):
    """Columns that start 8 bytes off a 16-byte boundary (a slice of a larger array) must not be touched with 16-byte
    accesses: also the caller's velocity columns, which the carried implicit-midpoint rounds read and write through
    the integrator's own pointers.  Same result as with aligned columns, bit for bit."""
    import torch
    n = 4099
    cols, pp, _ = synthetic.disk(n, 2)
    ops = build_ops(C5_STACK, pp)
    out = []
    for shifted in (False, True):
        dp = device_state(cols, pp)
        if shifted:
            for k in odd:
                room = torch.zeros(n + 3, dtype=torch.float64, device="cuda:0")
                start = 1 if room.data_ptr() % 16 == 0 else 2
                view = room[start:start + n]
                assert view.data_ptr() % 16 == 8
                view.copy_(dp.cols[k])
                dp.cols[k] = view
        rounds = ctx.integrate_force(dp.view(), ops, integrator, 0.01, EXACT)
        out.append((rounds, vel(dp), np.array(dp.acc_host())))
    assert out[0][0] == out[1][0]
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    ref = oracle.Sim(cols)
    oracle.integrate_force(ref, integrator, 0.01, C5_STACK, pp)
    v0 = np.array([cols[k] for k in ("vx", "vy", "vz")])
    assert dv_err(v0, np.array([ref.cols[k] for k in ("vx", "vy", "vz")]), out[1][1]) <= TOL


@pytest.mark.parametrize("dt", [1e-6, 0.01, 0.5, 2.0])
def test_implicit_midpoint_rounds_queued_ahead_of_the_host(ctx, dt):
    """The carried implicit-midpoint update tests convergence on the device and queues rounds ahead
    of the host (four, then two at a time): whether the test is met inside the first batch (the
    rounds behind it must then leave everything alone), at its end, or in a later batch, the
    number of rounds and the velocities are those of the separate passes, bit for bit in
    reference-order arithmetic, and the count is the checker's (these steps leave the loop at
    index 2, 3, 5 and 6)."""
    n = 3001
    cols, pp, _ = synthetic.disk(n, 2)
    ops = build_ops(C5_STACK, pp)
    out, rounds = [], []
    for carried in (True, False):
        dp = device_state(cols, pp)
        try:
            ctx.carried_steps(carried)
            rounds.append(ctx.integrate_force(dp.view(), ops, "implicit_midpoint", dt, EXACT))
        finally:
            ctx.carried_steps(True)
        out.append(vel(dp))
    assert rounds[0] == rounds[1]
    assert np.array_equal(out[0], out[1])
    ref = oracle.Sim(cols)
    assert oracle.integrate_force(ref, "implicit_midpoint", dt, C5_STACK, pp) == rounds[0]


@pytest.mark.parametrize("integrator", ["rk4", "implicit_midpoint"])
def test_every_particle_massive_uses_reductions(ctx, integrator):
    n = 20011
    cols, pp, _ = synthetic.disk(n, 1, dust_mass=1e-12)
    pp["primary"] = np.zeros(n, dtype=np.uint8)
    pp["primary"][0] = 1
    forces = [("tides_precession", {}), ("modify_orbits_forces", {"coordinates": 2}), ("gr_potential", {"c": C})]
    v0, vr, vd, *_ = run_pair(ctx, cols, pp, forces, integrator, 0.02, EXACT)
    assert dv_err(v0[:, 1:], vr[:, 1:], vd[:, 1:]) <= TOL
    # the star's update is a sum over all bodies
    push = np.abs(vr[:, 1:] - v0[:, 1:]).max() * 1e-12 * n
    assert np.abs(vd[:, 0] - vr[:, 0]).max() <= 1e-12 * max(push, np.abs(vr[:, 0] - v0[:, 0]).max()) + 4e-16 * np.abs(vr[:, 0]).max()


def test_none_integrator_only_clears_accelerations(ctx):
    cols, pp, _ = synthetic.disk(100, 1)
    synthetic.add_newtonian(cols, n_head=2)
    dp = device_state(cols, pp)
    ctx.integrate_force(dp.view(), build_ops(C5_STACK, pp), "none", 0.1)
    assert np.array_equal(vel(dp), np.array([cols[k] for k in ("vx", "vy", "vz")]))
    assert not np.array(dp.acc_host()).any()


def test_implicit_midpoint_reports_non_convergence(ctx):
    """A drag so strong that the fixed-point iteration diverges: 10 rounds, warning bit 4
    (reference: reb_warning, integrator_implicit_midpoint.c:116-118)."""
    n = 50
    cols, pp, _ = synthetic.disk(n, 1)
    pp["tau_a"] = np.full(n, np.nan)
    pp["tau_a"][1:] = -1e-3
    forces = [("modify_orbits_forces", {})]
    ref = oracle.Sim(cols)
    assert oracle.integrate_force(ref, "implicit_midpoint", 1.0, forces, pp) == 10
    dp = device_state(cols, pp)
    ctx.fetch_warnings()
    assert ctx.integrate_force(dp.view(), build_ops(forces, pp), "implicit_midpoint", 1.0, EXACT) == 10
    assert ctx.fetch_warnings() & 4


@pytest.mark.parametrize("mode", [EXACT, FAST])
def test_kick_and_drift(ctx, mode):
    n = 4097
    cols, pp, _ = synthetic.disk(n, 1)
    synthetic.add_newtonian(cols, n_head=2)
    ref = oracle.Sim(cols)
    oracle.kick_step(ref, 0.03)
    oracle.drift_step(ref, 0.07)
    dp = device_state(cols, pp)
    ctx.kick(dp.view(), 0.03, mode)
    ctx.drift(dp.view(), 0.07, mode)
    got = np.array([dp.cols[k].cpu().numpy() for k in ("x", "y", "z", "vx", "vy", "vz")])
    want = np.array([ref.cols[k] for k in ("x", "y", "z", "vx", "vy", "vz")])
    if mode == EXACT:
        assert ulp_diff(got, want).max() == 0
    else:
        # fused multiply-add: one rounding instead of two; where x + dt*v cancels, the
        # difference is bounded by the size of the terms, not of the result
        vk = np.array([cols[k] for k in ("vx", "vy", "vz")]) + 0.03 * np.array([cols[k] for k in ("ax", "ay", "az")])
        terms = np.concatenate([np.abs(np.array([cols[k] for k in ("x", "y", "z")])) + 0.07 * np.abs(vk), np.abs(vk) + 1.0])
        assert (np.abs(got - want) <= 2.3e-16 * terms).all()


@pytest.mark.parametrize("n", [0, 1, 2])
def test_smallest_particle_counts_go_through_every_stepping_entry_point(ctx, n):
    """Empty arrays, a lone star, star + one grain: no faults, and (n = 2) the checker's numbers."""
    import torch
    from reboundx_amd._lib import OP_GR_POTENTIAL, Op
    cols, pp, _ = synthetic.disk(max(n, 1), 0)
    cols = {k: v[:n] for k, v in cols.items()}
    pp = {k: v[:n] for k, v in pp.items()}
    dp = device_state(cols, pp, n_head=1, tail_massless=True) if n else None
    if n == 0:
        from reboundx_amd.device import DeviceParticles
        dp = DeviceParticles.from_host({k: np.zeros(0) for k in ("x", "y", "z", "vx", "vy", "vz", "m", "ax", "ay", "az")}, device="cuda:0")
    ops = build_ops(C5_STACK, pp) if n else []
    for integ in ("euler", "rk2", "rk4", "implicit_midpoint"):
        ctx.integrate_force(dp.view(), ops, integ, 0.01)
    ctx.kick(dp.view(), 0.01)
    ctx.drift(dp.view(), 0.01)
    if n >= 1:
        ctx.wh_drift(dp.view(), 0.01)
        ctx.wh_steps(dp.view(), ops, 0.01, 3)
        assert ctx.potentials(dp.view(), [Op(OP_GR_POTENTIAL, 0, 0, 0, C, 0.)])[0] <= 0.0
        assert np.isfinite(ctx.gr_hamiltonian(dp.view(), C))
        ctx.modify_mass(dp.view(), torch.full((n,), float("nan"), dtype=torch.float64, device="cuda:0"), 0.1)
        ctx.modify_orbits_direct(dp.view(), 0.1)
        got = np.array([dp.cols[k].cpu().numpy() for k in ("x", "y", "z", "vx", "vy", "vz")])
        assert np.isfinite(got).all()
    torch.cuda.synchronize()


@pytest.mark.parametrize("integrator", ["euler", "rk4", "implicit_midpoint"])
def test_grids_large_enough_for_the_xcd_tile_order(ctx, integrator):
    """From 4096 tiles of 512 bodies on, the streaming pass hands each XCD runs of consecutive tiles instead of tile = blockIdx,
    and pads its grid to whole groups of runs.  Everything a workgroup leaves behind is indexed by its tile: the carried update
    equals the separate passes bit for bit (both take the mapped order in their force passes; the partial sums of the
    implicit scheme's convergence test are per tile), and the explicit scheme equals the checker."""
    n = 2_150_003                       # 4200 tiles: the last group of runs is mostly padding, the last tile partial
    cols, pp, _ = synthetic.disk(n, 1)
    ops = build_ops(C5_STACK, pp)
    out, rounds = [], []
    for carried in (True, False):
        dp = device_state(cols, pp)
        try:
            ctx.carried_steps(carried)
            rounds.append(ctx.integrate_force(dp.view(), ops, integrator, 0.01, EXACT))
        finally:
            ctx.carried_steps(True)
        out.append((vel(dp), np.array(dp.acc_host())))
        del dp
    assert rounds[0] == rounds[1]
    assert np.isfinite(out[0][0]).all()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    if integrator == "euler":
        ref = oracle.Sim(cols)
        oracle.integrate_force(ref, integrator, 0.01, C5_STACK, pp)
        vr = np.array([ref.cols[k] for k in ("vx", "vy", "vz")])
        assert ulp_diff(out[0][0], vr).max() == 0
