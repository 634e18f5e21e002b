"""Device-resident Wisdom-Holman stepping (rebxamd_wh_steps: Kepler drift about the dominant
mass, additional forces in the kick) against the checker's Wisdom-Holman restatement
(oracle/rebx_whfast.c), which itself reproduces the reference's trajectory archives.

Layout: star + massless dust (the debris-disk configuration).  The two sides differ in how
they cancel the star's pull on bodies behind the first (the checker adds direct gravity and the
Jacobi correction, which cancel to rounding; the device never forms them), so agreement is to
rounding accumulated over the steps, not bitwise: bar 1e-12 of the state after 40 steps."""
import numpy as np
import pytest
import torch

import oracle
from reboundx_amd import synthetic
from reboundx_amd.device import DeviceParticles
from reboundx_amd.program import build_ops

pytestmark = pytest.mark.gpu
C = synthetic.C_DEFAULT
C5_STACK = [("radiation_forces", {"c": C}), ("gravitational_harmonics", {}), ("gr_potential", {"c": C}),
            ("tides_precession", {}), ("modify_orbits_forces", {})]


@pytest.fixture(scope="module")
def ctx():
    from reboundx_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def state_of(dp):
    return np.array([dp.cols[k].cpu().numpy() for k in ("x", "y", "z", "vx", "vy", "vz")])


def device_particles(cols, pp):
    host = dict(cols)
    for k in ("beta", "R_tides", "k1", "tau_a", "tau_e", "tau_inc"):
        host[k] = pp[k]
    return DeviceParticles.from_host(host, device="cuda:0", n_head=1, tail_massless=True)


@pytest.mark.parametrize("mode", [1, 0])
@pytest.mark.parametrize("forces", [[], C5_STACK, [("gr", {"c": C})]])
def test_steps_match_the_checker(ctx, forces, mode):
    n, dt, steps = 1500, 0.05, 40
    cols, pp, _ = synthetic.disk(n, 0)
    cols["vx"][0], cols["vy"][0] = 1e-3, -2e-3          # a moving star exercises the linear drift
    ref = oracle.Sim(cols)
    oracle.whfast_steps(ref, dt, steps, forces, pp)
    want = np.array([ref.cols[k] for k in ("x", "y", "z", "vx", "vy", "vz")])
    dp = device_particles(cols, pp)
    ctx.wh_steps(dp.view(), build_ops(forces, pp), dt, steps, mode)
    got = state_of(dp)
    assert np.isfinite(got).all()
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= 1e-12 * scale


C2_STACK = [("radiation_forces", {"c": C}), ("gravitational_harmonics", {}), ("gr", {"c": C})]


@pytest.mark.parametrize("mode", [1, 0])
@pytest.mark.parametrize("forces", [C5_STACK, C2_STACK, [("gr", {"c": C})], [("modify_orbits_forces", {"coordinates": 1}), ("gr_potential", {"c": C})]],
                         ids=["five-effects", "rad+harmonics+gr", "gr", "barycentric-drag+gr_potential"])
@pytest.mark.parametrize("n", [2, 3, 1500, 4099])
def test_steps_carried_by_the_force_pass_equal_separate_passes(ctx, n, forces, mode):
    """rebxamd_wh_steps lets the pass over the massless bodies carry kick and drift; run with that
    switched off, force pass and kick+drift pass are separate kernels.  Same expressions: in
    reference-order arithmetic the states must agree bit for bit, and the acceleration columns
    are zero afterwards either way."""
    dt, steps = 0.05, 7
    cols, pp, _ = synthetic.disk(n, 0)
    cols["vx"][0], cols["vy"][0] = 1e-3, -2e-3
    ops = build_ops(forces, pp)
    out = []
    for carried in (True, False):
        dp = device_particles(cols, pp)
        try:
            ctx.carried_steps(carried)
            ctx.wh_steps(dp.view(), ops, dt, steps, mode)
        finally:
            ctx.carried_steps(True)
        out.append(state_of(dp))
        assert float(np.abs(np.array(dp.acc_host())).max()) == 0.0
    assert np.isfinite(out[0]).all()
    if mode == 1:
        assert np.array_equal(out[0], out[1])
    else:
        assert np.abs(out[0] - out[1]).max() <= 1e-13 * np.abs(out[1]).max()


def test_acceleration_columns_are_scratch_for_the_stepping_run(ctx):
    """Whatever the acceleration columns hold on entry has no influence on the run: body 0's acceleration starts from
    zero inside the head pass, the forces on the other bodies never leave registers."""
    n, dt, steps = 3001, 0.05, 5
    cols, pp, _ = synthetic.disk(n, 0)
    ops = build_ops(C5_STACK, pp)
    out = []
    for garbage in (False, True):
        c = dict(cols)
        if garbage:
            rng = np.random.default_rng(1)
            for k in ("ax", "ay", "az"):
                c[k] = rng.normal(size=n) * 1e3
        dp = device_particles(c, pp)
        ctx.wh_steps(dp.view(), ops, dt, steps, 1)
        out.append(state_of(dp))
    assert np.array_equal(out[0], out[1])


def eccentric_swarm(n, seed=11):
    """Star + massless bodies with e up to 0.95, all inclinations, a in [0.5, 3]."""
    rng = np.random.default_rng(seed)
    a, e = rng.uniform(0.5, 3.0, n - 1), rng.uniform(0.0, 0.95, n - 1)
    inc, Om, om, f = rng.uniform(0, np.pi, n - 1), rng.uniform(0, 2 * np.pi, n - 1), rng.uniform(0, 2 * np.pi, n - 1), rng.uniform(0, 2 * np.pi, n - 1)
    vals = synthetic.kepler_to_cartesian(1.0, a, e, inc, Om, om, f)
    cols = {k: np.concatenate([[0.0], v]) for k, v in zip(("x", "y", "z", "vx", "vy", "vz"), vals)}
    cols["m"] = np.zeros(n)
    cols["m"][0] = 1.0
    for k in ("ax", "ay", "az"):
        cols[k] = np.zeros(n)
    return cols


@pytest.mark.parametrize("mode", [1, 0])
@pytest.mark.parametrize("population", ["disk", "eccentric"])
def test_kepler_drift_conserves_the_orbit_and_closes_after_one_period(ctx, population, mode):
    """Independent of the checker: energy and angular momentum of each body about the star are
    unchanged by a drift, a drift of -dt undoes a drift of dt, and a drift by the orbital period
    returns every body to where it started.  Tolerances scale with 1/(1-e): near pericentre of
    an eccentric orbit a rounding error in time is a large error in position."""
    n = 3000
    if population == "disk":
        cols, pp, _ = synthetic.disk(n, 0)
        dp = device_particles(cols, pp)
        amp = 1.0
    else:
        cols = eccentric_swarm(n)
        dp = DeviceParticles.from_host(cols, device="cuda:0", n_head=1, tail_massless=True)
        amp = 400.0
    s0 = state_of(dp)

    def invariants(s):
        r = s[:3, 1:] - s[:3, :1]
        v = s[3:, 1:] - s[3:, :1]
        energy = 0.5 * (v ** 2).sum(axis=0) - 1.0 / np.sqrt((r ** 2).sum(axis=0))
        return energy, np.cross(r.T, v.T).T

    e0, l0 = invariants(s0)
    ctx.wh_drift(dp.view(), 0.37, mode)
    s1 = state_of(dp)
    e1, l1 = invariants(s1)
    assert np.abs((e1 - e0) / e0).max() <= 1e-14 * amp
    assert np.abs(l1 - l0).max() <= 1e-14 * amp
    assert np.abs(s1 - s0).max() > 0.1
    ctx.wh_drift(dp.view(), -0.37, mode)
    assert np.abs(state_of(dp) - s0).max() <= 1e-13 * amp
    # one full period of a chosen body
    a = -0.5 / e0
    period = 2 * np.pi * a ** 1.5
    k = 17
    ctx.wh_drift(dp.view(), float(period[k]), mode)
    s2 = state_of(dp)
    assert np.abs(s2[:, k + 1] - s0[:, k + 1]).max() <= 1e-12 * amp


def test_kepler_drift_default_arithmetic_agrees_with_reference_order(ctx):
    cols = eccentric_swarm(5000, seed=3)
    out = []
    for mode in (1, 0):
        dp = DeviceParticles.from_host(cols, device="cuda:0", n_head=1, tail_massless=True)
        for _ in range(10):
            ctx.wh_drift(dp.view(), 0.21, mode)
        out.append(state_of(dp))
    r = np.sqrt((cols["x"] ** 2 + cols["y"] ** 2 + cols["z"] ** 2))[1:]
    assert np.abs(out[0] - out[1])[:, 1:].max() <= 1e-11        # eccentric orbits amplify time rounding near pericentre
    assert np.isfinite(out[1]).all() and r.min() > 0


@pytest.mark.parametrize("mode", [1, 0])
@pytest.mark.parametrize("forces", [[], C5_STACK, C2_STACK, [("gr", {"c": C})], [("modify_orbits_forces", {"coordinates": 1}), ("gr_potential", {"c": C})]],
                         ids=["force-free", "five-effects", "rad+harmonics+gr", "gr", "barycentric-drag+gr_potential"])
@pytest.mark.parametrize("n,n_planets", [(3, 2), (1500, 2), (700, 6)])
def test_steps_of_star_planets_and_dust_match_the_checker(ctx, n, n_planets, forces, mode):
    """BASELINE configuration 1's shape (star + planets + massless dust) stepped on the device: the massive head in Jacobi
    coordinates in one wavefront (Kepler drifts about G*eta_i, gravity among the planets without the star-planet-1 pair,
    Jacobi kick), the dust behind it in one pass per step about the head's centre of mass, pulled by every head body.
    Against the checker's Wisdom-Holman restatement (which reproduces the reference's two-planet archives), 40 steps.
    The two sides round differently (whole drifts between kicks here, two half drifts with a change of coordinates in
    between there; the star's pull on a dust grain and the Jacobi term cancel to 1e-3 of either): bar 1e-11 of the state."""
    dt, steps = 0.05, 40
    cols, pp, _ = synthetic.disk(n, n_planets)
    cols["vx"][0], cols["vy"][0] = 1e-3, -2e-3
    ref = oracle.Sim(cols)
    oracle.whfast_steps(ref, dt, steps, forces, pp)
    want = np.array([ref.cols[k] for k in ("x", "y", "z", "vx", "vy", "vz")])
    host = dict(cols)
    for k in ("beta", "R_tides", "k1", "tau_a", "tau_e", "tau_inc"):
        host[k] = pp[k]
    dp = DeviceParticles.from_host(host, device="cuda:0", n_head=n_planets + 1, tail_massless=True)
    ctx.wh_steps(dp.view(), build_ops(forces, pp), dt, steps, mode)
    got = state_of(dp)
    assert np.isfinite(got).all()
    start = np.array([cols[k] for k in ("x", "y", "z", "vx", "vy", "vz")])
    assert np.abs(want - start).max() > 0.1
    err = np.abs(got - want).max(axis=0)
    scale = np.abs(want).max()
    from gpu_common import record
    record("wh_steps:star+%d planets:n=%d:%s:mode=%d" % (n_planets, n, "+".join(f[0] for f in forces) or "force-free", mode), unfloored=err.max() / scale)
    assert err.max() <= 1e-11 * scale, (int(err.argmax()), err.max() / scale)
    assert float(np.abs(np.array(dp.acc_host())[:, :n_planets + 1]).max()) == 0.0


@pytest.mark.parametrize("mode", [1, 0])
@pytest.mark.parametrize("n,n_planets", [(20_011, 3), (9, 7), (2500, 1)])
def test_head_step_riding_in_the_pass_changes_nothing(ctx, n, n_planets, mode):
    """After the first step, the head pass and the head's own step of step k + 1 run in an extra workgroup of the pass
    over the massless bodies of step k (they never read a massless body), into a second copy of the tables and of the
    record.  Switched off, every step launches them ahead of its pass.  Same code, same inputs: the same bits."""
    dt, steps = 0.05, 9
    cols, pp, _ = synthetic.disk(n, n_planets)
    host = dict(cols)
    for k in ("beta", "R_tides", "k1", "tau_a", "tau_e", "tau_inc"):
        host[k] = pp[k]
    ops = build_ops(C5_STACK, pp)
    out = []
    for riding in (True, False):
        dp = DeviceParticles.from_host(host, device="cuda:0", n_head=n_planets + 1, tail_massless=True)
        try:
            ctx.carried_steps(riding)
            ctx.wh_steps(dp.view(), ops, dt, steps, mode)
        finally:
            ctx.carried_steps(True)
        out.append(state_of(dp))
    assert np.isfinite(out[0]).all()
    assert np.array_equal(out[0], out[1])


def test_star_and_planets_conserve_energy_and_momentum_over_a_long_run(ctx):
    """Independent of the checker: force-free, 2000 steps of star + 4 planets: total momentum to rounding (the centre of
    mass is never kicked), energy error bounded and oscillating as for any Wisdom-Holman map (< 1e-7 relative here, with
    no secular drift between the two halves of the run), massless grains keep their Jacobi constant to the same level."""
    n, n_planets, dt = 300, 4, 0.02
    cols, pp, _ = synthetic.disk(n, n_planets)

    def diagnostics(s):
        m = cols["m"][:n_planets + 1]
        x, v = s[:3, :n_planets + 1], s[3:, :n_planets + 1]
        kinetic = 0.5 * (m * (v ** 2).sum(axis=0)).sum()
        pot = 0.
        for i in range(n_planets + 1):
            for j in range(i):
                pot -= m[i] * m[j] / np.sqrt(((x[:, i] - x[:, j]) ** 2).sum())
        return kinetic + pot, (m * v).sum(axis=1)

    dp = DeviceParticles.from_host(dict(cols), device="cuda:0", n_head=n_planets + 1, tail_massless=True)
    e0, p0 = diagnostics(state_of(dp))
    errs = []
    for _ in range(2):
        ctx.wh_steps(dp.view(), [], dt, 1000, 0)
        e, p = diagnostics(state_of(dp))
        errs.append(abs((e - e0) / e0))
        assert np.abs(p - p0).max() <= 1e-15
    assert max(errs) <= 1e-7 and errs[1] <= 10 * errs[0] + 1e-9
    assert np.isfinite(state_of(dp)).all()


def test_layouts_the_stepping_run_cannot_take_are_refused(ctx):
    from reboundx_amd.device import RebxAmdError
    cols, pp, _ = synthetic.disk(100, 2, dust_mass=1e-9)         # every body massive: more than 64 in the head
    dp = DeviceParticles.from_host(dict(cols), device="cuda:0")
    with pytest.raises(RebxAmdError) as ei:
        ctx.wh_steps(dp.view(), [], 0.01, 1)
    assert ei.value.code == 3
    cols, pp, _ = synthetic.disk(100, 2)
    dp = DeviceParticles.from_host(dict(cols), device="cuda:0", n_head=3, tail_massless=True)
    with pytest.raises(RebxAmdError) as ei:                      # gr_full is not a streaming force
        ctx.wh_steps(dp.view(), build_ops([("gr_full", {"c": C})], pp), 0.01, 1)
    assert ei.value.code == 3


@pytest.mark.parametrize("mode", [1, 0])
@pytest.mark.parametrize("forces", [C5_STACK, C2_STACK, [("radiation_forces", {"c": C})]],
                         ids=["five-effects", "rad+harmonics+gr", "radiation"])
@pytest.mark.parametrize("n", [2, 3, 1500, 4099])
def test_run_left_open_between_calls_ends_bit_for_bit_where_the_closed_run_ends(ctx, n, forces, mode):
    """rebxamd_wh_run_begin / _steps / _end: the trailing half drift of a step waits for the next pass (or for the end of the
    run).  The operations on every body are those of rebxamd_wh_steps in the same order, so after the closing call the states
    are equal to the bit in both arithmetic modes, however the steps were split over calls; the host may wait after any call."""
    dt = 0.05
    cols, pp, _ = synthetic.disk(n, 0)
    cols["vx"][0], cols["vy"][0] = 1e-3, -2e-3
    ops = build_ops(forces, pp)
    closed = device_particles(cols, pp)
    ctx.wh_steps(closed.view(), ops, dt, 7, mode)
    want = state_of(closed)
    for split in ([7], [1] * 7, [3, 0, 4], [1, 5, 1]):
        dp = device_particles(cols, pp)
        ctx.wh_run_begin(dp.view(), ops, dt, mode)
        for k in split:
            ctx.wh_run_steps(k)
            torch.cuda.synchronize()
        ctx.wh_run_end()
        got = state_of(dp)
        assert np.isfinite(got).all()
        assert np.array_equal(got, want), split


@pytest.mark.parametrize("mode", [1, 0])
@pytest.mark.parametrize("forces", [[], C5_STACK, [("radiation_forces", {"c": C})], [("modify_orbits_forces", {"coordinates": 1}), ("gr_potential", {"c": C})]],
                         ids=["force-free", "five-effects", "radiation", "barycentric-drag+gr_potential"])
@pytest.mark.parametrize("n,n_planets", [(4, 2), (1500, 1), (4099, 6)])
def test_run_left_open_behind_star_and_planets_ends_bit_for_bit_where_the_closed_run_ends(ctx, n, n_planets, forces, mode):
    """The same behind a massive head of several bodies: every pass drifts the massless bodies to its force evaluation about
    the frame record of the step before and kicks them; the head, one step ahead in the extra workgroup, keeps the state it had
    between its kick and its drift, from which the closing call drifts it half a step.  Same operations in the same order as
    rebxamd_wh_steps: equal to the bit in both arithmetic modes for every split of the steps over calls."""
    dt = 0.05
    cols, pp, _ = synthetic.disk(n, n_planets)
    cols["vx"][0], cols["vy"][0] = 1e-3, -2e-3
    host = dict(cols)
    for k in ("beta", "R_tides", "k1", "tau_a", "tau_e", "tau_inc"):
        host[k] = pp[k]
    ops = build_ops(forces, pp)
    closed = DeviceParticles.from_host(host, device="cuda:0", n_head=n_planets + 1, tail_massless=True)
    ctx.wh_steps(closed.view(), ops, dt, 7, mode)
    want = state_of(closed)
    start = np.array([cols[k] for k in ("x", "y", "z", "vx", "vy", "vz")])
    assert np.abs(want - start).max() > 0.1
    for split in ([7], [1] * 7, [3, 0, 4], [1, 5, 1], [2, 2, 3]):
        dp = DeviceParticles.from_host(host, device="cuda:0", n_head=n_planets + 1, tail_massless=True)
        ctx.wh_run_begin(dp.view(), ops, dt, mode)
        for k in split:
            ctx.wh_run_steps(k)
            torch.cuda.synchronize()
        ctx.wh_run_end()
        got = state_of(dp)
        assert np.isfinite(got).all()
        assert np.array_equal(got, want), (split, float(np.abs(got - want).max()))
        assert float(np.abs(np.array(dp.acc_host())[:, :n_planets + 1]).max()) == 0.0
    # a run that took no step leaves the columns untouched
    dp = DeviceParticles.from_host(host, device="cuda:0", n_head=n_planets + 1, tail_massless=True)
    before = state_of(dp)
    ctx.wh_run_begin(dp.view(), ops, dt, mode)
    ctx.wh_run_end()
    assert np.array_equal(state_of(dp), before)


def test_open_run_contract(ctx):
    """A run that took no step leaves the columns untouched; a second begin, steps or end without a run, and layouts or
    stacks the open run does not take are refused; after the end the context steps as before."""
    from reboundx_amd.device import RebxAmdError
    cols, pp, _ = synthetic.disk(500, 0)
    ops = build_ops(C5_STACK, pp)
    dp = device_particles(cols, pp)
    before = state_of(dp)
    ctx.wh_run_begin(dp.view(), ops, 0.05)
    with pytest.raises(RebxAmdError) as ei:
        ctx.wh_run_begin(dp.view(), ops, 0.05)
    assert ei.value.code == 1
    ctx.wh_run_steps(0)
    ctx.wh_run_end()
    assert np.array_equal(state_of(dp), before)
    for call in (lambda: ctx.wh_run_steps(1), ctx.wh_run_end):
        with pytest.raises(RebxAmdError) as ei:
            call()
        assert ei.value.code == 1
    planets, ppp, _ = synthetic.disk(100, 2)
    dq = DeviceParticles.from_host(dict(planets), device="cuda:0", n_head=3, tail_massless=True)
    for view, stack, par in ((dq.view(), [("gr_full", {"c": C})], ppp), (dp.view(), [], pp), (dp.view(), [("modify_orbits_forces", {"coordinates": 1})], pp),
                             (dp.view(), [("gr_full", {"c": C})], pp)):
        with pytest.raises(RebxAmdError) as ei:
            ctx.wh_run_begin(view, build_ops(stack, par), 0.05)
        assert ei.value.code == 3
    ctx.wh_steps(dp.view(), ops, 0.05, 2)                        # no run is left open by a refused begin
    assert np.isfinite(state_of(dp)).all()


@pytest.mark.parametrize("n_planets", [0, 2])
def test_open_run_keeps_its_tables_to_itself(ctx, n_planets):
    """While a run is open, the entry points that would overwrite the context's tables and records are refused (and leave
    nothing behind: the run ends bit for bit where an undisturbed one ends), the run stays on its stream, and a second
    context serves the same calls meanwhile."""
    import torch
    from reboundx_amd.device import Context, RebxAmdError
    cols, pp, _ = synthetic.disk(3000, n_planets)
    stack = [("radiation_forces", {"c": C})] if n_planets else C5_STACK
    ops = build_ops(stack, pp)
    kw = dict(n_head=1 + n_planets, tail_massless=True)
    host = dict(cols)
    for k in ("beta", "R_tides", "k1", "tau_a", "tau_e", "tau_inc"):
        host[k] = pp[k]
    quiet = DeviceParticles.from_host(host, device="cuda:0", **kw)
    ctx.wh_run_begin(quiet.view(), ops, 0.03)
    ctx.wh_run_steps(5)
    ctx.wh_run_end()
    dp = DeviceParticles.from_host(host, device="cuda:0", **kw)
    other = DeviceParticles.from_host(host, device="cuda:0", **kw)
    ctx.wh_run_begin(dp.view(), ops, 0.03)
    ctx.wh_run_steps(2)
    for call in (lambda: ctx.apply(other.view(), ops, 0), lambda: ctx.wh_steps(other.view(), ops, 0.03, 1), lambda: ctx.wh_drift(other.view(), 0.03),
                 lambda: ctx.integrate_force(other.view(), ops, "rk2", 0.01), lambda: ctx.wh_batch(4)):
        with pytest.raises(RebxAmdError) as ei:
            call()
        assert ei.value.code == 1 and "run is open" in str(ei.value)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for call in (lambda: ctx.wh_run_steps(1), ctx.wh_run_end):
            with pytest.raises(RebxAmdError) as ei:
                call()
            assert ei.value.code == 1 and "stream" in str(ei.value)
    second = Context(0)
    second.apply(other.view(), ops, 0)                            # a second context is free to work meanwhile
    second.close()
    ctx.wh_run_steps(3)
    ctx.wh_run_end()
    assert np.array_equal(state_of(dp), state_of(quiet))


@pytest.mark.parametrize("mode", [1, 0])
@pytest.mark.parametrize("forces", [C5_STACK, C2_STACK, [("gr", {"c": C})]], ids=["five-effects", "rad+harmonics+gr", "gr"])
@pytest.mark.parametrize("n", [2, 3, 1500, 4099])
def test_batches_of_steps_in_one_pass_equal_one_pass_per_step(ctx, n, forces, mode):
    """Behind a single massive body every massless body can go through a batch of steps in registers (rebxamd_wh_batch).  The
    operations on each body are those of one pass per step, in the same order, with the same tables: the states agree to the
    bit in both arithmetic modes, whatever the batch length and however the run length falls on it."""
    dt = 0.05
    cols, pp, _ = synthetic.disk(n, 0)
    cols["vx"][0], cols["vy"][0] = 1e-3, -2e-3
    ops = build_ops(forces, pp)
    try:
        for steps in (1, 2, 5, 8, 9, 13):
            ctx.wh_batch(1)
            one = device_particles(cols, pp)
            ctx.wh_steps(one.view(), ops, dt, steps, mode)
            want = state_of(one)
            for batch in (2, 4, 32):
                ctx.wh_batch(batch)
                dp = device_particles(cols, pp)
                ctx.wh_steps(dp.view(), ops, dt, steps, mode)
                got = state_of(dp)
                assert np.isfinite(got).all()
                assert np.array_equal(got, want), (steps, batch)
    finally:
        ctx.wh_batch(32)


@pytest.mark.parametrize("mode", [1, 0])
@pytest.mark.parametrize("forces", [[], C5_STACK, [("radiation_forces", {"c": C})], [("modify_orbits_forces", {"coordinates": 1}), ("gr_potential", {"c": C})]],
                         ids=["force-free", "five-effects", "radiation", "barycentric-drag+gr_potential"])
@pytest.mark.parametrize("n,n_planets", [(4, 2), (1500, 1), (4099, 6)])
def test_batches_of_steps_behind_star_and_planets_equal_one_pass_per_step(ctx, n, n_planets, forces, mode):
    """The same behind a head of several massive bodies: the head goes through the steps of a batch first (one wavefront, the
    tables and the frame record of every step kept), then one pass takes every massless body through them.  Same operations
    per body as one pass per step: the same bits, head bodies included."""
    dt = 0.05
    cols, pp, _ = synthetic.disk(n, n_planets)
    cols["vx"][0], cols["vy"][0] = 1e-3, -2e-3
    host = dict(cols)
    for k in ("beta", "R_tides", "k1", "tau_a", "tau_e", "tau_inc"):
        host[k] = pp[k]
    ops = build_ops(forces, pp)
    try:
        for steps in (1, 2, 7, 9):
            ctx.wh_batch(1)
            one = DeviceParticles.from_host(host, device="cuda:0", n_head=n_planets + 1, tail_massless=True)
            ctx.wh_steps(one.view(), ops, dt, steps, mode)
            want = state_of(one)
            for batch in (2, 4, 32):
                ctx.wh_batch(batch)
                dp = DeviceParticles.from_host(host, device="cuda:0", n_head=n_planets + 1, tail_massless=True)
                ctx.wh_steps(dp.view(), ops, dt, steps, mode)
                got = state_of(dp)
                assert np.isfinite(got).all()
                assert np.array_equal(got, want), (steps, batch)
                assert float(np.abs(np.array(dp.acc_host())[:, :n_planets + 1]).max()) == 0.0
    finally:
        ctx.wh_batch(32)


@pytest.mark.parametrize("n_planets", [0, 2])
def test_batches_on_a_grid_large_enough_for_the_xcd_tile_order(ctx, n_planets):
    """The same comparison where the pass hands each XCD runs of consecutive tiles (from 4096 tiles of 512 bodies on) and has
    its extra workgroup in front of a padded grid."""
    n, dt, steps = 2_150_003, 0.05, 7
    cols, pp, _ = synthetic.disk(n, n_planets)
    host = dict(cols)
    for k in ("beta", "R_tides", "k1", "tau_a", "tau_e", "tau_inc"):
        host[k] = pp[k]
    ops = build_ops(C5_STACK, pp)
    out = []
    try:
        for batch in (1, 3):
            ctx.wh_batch(batch)
            dp = DeviceParticles.from_host(host, device="cuda:0", n_head=n_planets + 1, tail_massless=True)
            ctx.wh_steps(dp.view(), ops, dt, steps, 1)
            out.append(state_of(dp))
            del dp
    finally:
        ctx.wh_batch(32)
    assert np.isfinite(out[0]).all()
    assert np.array_equal(out[0], out[1])
    start = np.array([cols[k] for k in ("x", "y", "z", "vx", "vy", "vz")])
    assert np.abs(out[0] - start).max() > 0.1


def sampled_run_deviation(ctx, n, steps, dt, mode, nsample=256, seed=5):
    """BASELINE configuration 5: star + n massless grains under the five stacked effects for `steps` Wisdom-Holman steps
    on the device (batches of steps per pass), and the same run of (star + a sample of grains) by the checker.  Massless
    grains feel the star and not one another, and nothing they do reaches the star, so the sample's trajectory is the same
    in both runs.  Returns max |device - checker| over the sample's positions and velocities, each grain's divided by
    max(1, its distance from the star)."""
    cols, pp, _ = synthetic.disk(n + 1, 0)
    sample = np.sort(np.concatenate([[0], 1 + np.random.default_rng(seed).choice(n, size=min(nsample, n), replace=False)]))
    dp = device_particles(cols, pp)
    ctx.wh_steps(dp.view(), build_ops(C5_STACK, {k: v[:1] for k, v in pp.items()}), dt, steps, mode)
    pick = torch.from_numpy(sample).to("cuda:0")
    got = np.array([dp.cols[k][pick].cpu().numpy() for k in ("x", "y", "z", "vx", "vy", "vz")])
    assert bool(torch.isfinite(dp.cols["x"]).all().item()) and bool(torch.isfinite(dp.cols["vx"]).all().item())
    ref = oracle.Sim({k: v[sample].copy() for k, v in cols.items()})
    oracle.whfast_steps(ref, dt, steps, C5_STACK, {k: v[sample].copy() for k, v in pp.items()})
    want = np.array([ref.cols[k] for k in ("x", "y", "z", "vx", "vy", "vz")])
    r = np.sqrt((want[:3] ** 2).sum(axis=0))
    return (np.abs(got - want).max(axis=0) / np.maximum(r, 1.)).max()


@pytest.mark.parametrize("mode", [0, 1])
def test_a_million_grains_over_ten_thousand_steps_end_where_the_checker_puts_a_sample(ctx, mode):
    """Configuration 5 at its real length: 1e6 grains x 1e4 steps of dt = P/100 (100 orbits), 256 sampled grains against
    the checker.  Bound: 2e-10 -- the 1e-12 of the 40-step tests grown linearly with the number of steps, as roundoff in
    the orbital phase grows (measured: 2.3e-11 in both arithmetic modes)."""
    dev = sampled_run_deviation(ctx, 1_000_000, 10_000, 2 * np.pi / 100.0, mode)
    assert dev <= 2e-10, dev
