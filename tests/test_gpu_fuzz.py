"""Randomised parity sweep of the streaming forces on the GPU against the CPU checker: random
stacks, mass layouts, unset parameters, extra sources and incoming accelerations, both
arithmetic modes and both kernel selections (instances built for a stack / walking the op
list).  Seeds are fixed: every run tests the same cases.

Bar: 1e-12 of the size of the particle's update vector (plus, for massive bodies, the size of
the pushes they collect from the others, which arrive as parallel sums).  Stacks that hold gr or
Jacobi/barycentric modify_orbits_forces on a layout with massive bodies beyond the head: in
reference-order arithmetic the device walks the reference's recurrences in the reference's order
(all sizes here are below its limit), so the same bar holds; in the default arithmetic those two
forces are parallel prefix sums, a different rounding of the same sums than the checker's
sequential one, and the bar gains the checker's own distance from the exact result, measured per
component without the device (|device - checker| <= |device - exact| + |checker - exact|): its
distance from the binary128 value of each such force evaluated on zero accelerations, plus what
the accelerations already in the columns do to its sequential additions -- a body with a large
acceleration that collects thousands of reactions each smaller than half an ulp of it loses every
one of them in the reference's order (seed 6956, body 1: the whole Jacobi reaction, 1.5e-15 on an
acceleration of 0.038) -- which is the checker's update in place minus its update from zero."""
import numpy as np
import pytest

from gpu_common import TOL, err_vs_exact, record, run_device, run_oracle, ulp_diff
from reboundx_amd import synthetic
from test_gpu_parity import ROUNDING, exact_and_sequential, is_prefix_sum_force, zero_acc

pytestmark = pytest.mark.gpu
C = synthetic.C_DEFAULT
FORCES = [("radiation_forces", {"c": C}), ("gravitational_harmonics", {}), ("gr", {"c": C}), ("gr", {"c": 200.0, "max_iterations": 5}),
          ("gr_potential", {"c": C}), ("central_force", {}), ("tides_precession", {}), ("modify_orbits_forces", {}),
          ("modify_orbits_forces", {"coordinates": 1}), ("modify_orbits_forces", {"coordinates": 2})]


@pytest.fixture(scope="module")
def ctx():
    from reboundx_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def random_case(seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([2, 3, 7, 130, 1025, 4100, 9001]))
    planets = int(rng.integers(0, min(5, n - 1) + 1))
    layout = rng.choice(["massless-tail", "all-massive", "scattered"])
    dust = {"massless-tail": 0.0, "all-massive": 10.0 ** rng.uniform(-14, -9), "scattered": 0.0}[layout]
    cols, pp, _ = synthetic.disk(n, planets, seed=int(rng.integers(1, 100)), dust_mass=dust)
    h = 1 + planets
    if layout == "scattered" and n > h + 2:
        for i in rng.choice(np.arange(h, n), size=min(4, n - h), replace=False):
            cols["m"][i] = 10.0 ** rng.uniform(-8, -4)
    # unset parameters here and there
    for key in ("beta", "tau_a", "tau_e", "tau_inc", "R_tides", "k1"):
        if key in pp and n > 4 and rng.random() < 0.6:
            pp[key] = pp[key].copy()
            pp[key][rng.choice(np.arange(1, n), size=max(1, n // 7), replace=False)] = np.nan
    pp["primary"] = np.zeros(n, dtype=np.uint8)
    pp["primary"][int(rng.integers(0, h))] = 1
    pp["Acentral"] = np.full(n, np.nan); pp["gammacentral"] = np.full(n, np.nan)
    pp["Acentral"][0], pp["gammacentral"][0] = 1e-4, float(rng.choice([-1.0, -2.0, 0.5]))
    if planets >= 1 and rng.random() < 0.5:                # a second oblate body / radiation source / tidal primary in the head
        for key, val in (("J2", 2e-3), ("J4", -1e-4), ("R_eq", 2e-3)):
            pp[key] = pp[key].copy(); pp[key][1] = val
        pp["radiation_source"] = np.zeros(n, dtype=np.uint8); pp["radiation_source"][[0, 1]] = 1
    if rng.random() < 0.5:
        synthetic.add_newtonian(cols, n_head=h)
    k = int(rng.integers(1, 6))
    picks = sorted(rng.choice(len(FORCES), size=k, replace=False))
    stack, seen_gr, seen_com = [], False, False
    for j in rng.permutation(picks):
        name, ep = FORCES[j]
        com = name == "modify_orbits_forces" and ep.get("coordinates", 0) in (0, 1)
        if (name == "gr" and seen_gr) or (com and seen_com):
            continue
        seen_gr |= name == "gr"
        seen_com |= com
        stack.append((name, ep))
    scan = layout != "massless-tail" and any(f[0] == "gr" or (f[0] == "modify_orbits_forces" and f[1].get("coordinates", 0) in (0, 1)) for f in stack)
    return cols, pp, stack, n, scan


@pytest.mark.parametrize("mode", [1, 0], ids=["exact", "fast"])
@pytest.mark.parametrize("seed", range(150))
def test_random_stack_layout_and_parameters(ctx, seed, mode):
    run_case(ctx, seed, mode)


def run_case(ctx, seed, mode):
    """Runs case `seed`; returns whether the binary128 term had to be added to the bar."""
    cols, pp, stack, n, scan = random_case(seed)
    ref = run_oracle(cols, pp, stack)
    base = np.stack([cols["ax"], cols["ay"], cols["az"]])
    upd = np.abs(ref - base).max(axis=0)
    m = cols["m"]
    # massive bodies collect pushes from the others: sum_j m_j |update_j| / m_i bounds their size
    total = float((m * upd).sum())
    react = np.where(m > 0, (total - m * upd) / np.where(m > 0, m, 1.0), 0.0)
    # floor: every force of the stack adds up to n terms one after the other into a_i in the reference (barycentric
    # reactions, pushes on a source), each addition rounding at the size of a_i; a regrouped sum differs by
    # ~sqrt(n x forces) such roundings
    den = np.maximum(upd + react, np.abs(ref).max(axis=0) * (4 + np.sqrt(n * len(stack))) * np.finfo(float).eps / TOL)
    den = np.where(den == 0.0, 1.0, den)
    bar = TOL * (16 if any(f[0] == "central_force" for f in stack) else 1)
    allowed = bar * den[None, :] * np.ones((3, 1))
    widened = False
    for fixed in (True, False):
        try:
            ctx.fixed_stacks(fixed)
            got, _ = run_device(ctx, cols, pp, stack, 1.0, mode)
        finally:
            ctx.fixed_stacks(True)
        assert np.isfinite(got).all(), (seed, stack)
        err = np.abs(got - ref)
        if scan and mode == 0 and not widened and (err > allowed).any():
            # (binary128 is software arithmetic: evaluated only for the cases that need it)
            for f in stack:
                if is_prefix_sum_force(f):
                    allowed = allowed + exact_and_sequential(cols, pp, f)[2]
            allowed = allowed + np.abs((ref - base) - run_oracle(zero_acc(cols), pp, stack))
            widened = True
        if mode == 1 and not any(f[0] == "central_force" for f in stack):
            # reference-order arithmetic: every body carries the checker's bits, on every mass layout -- the sources and
            # primaries too, which collect the pushes of all others one by one in the order the reference visits them
            # (central_force calls pow, which differs from the host's by a few ulp: not asserted for it)
            off = np.nonzero(ulp_diff(got, ref).max(axis=0))[0]
            assert off.size == 0, "seed %d stack %s: bodies %s differ from the checker's bits" % (seed, [f[0] for f in stack], off[:8].tolist())
        excess = (err - allowed) / den[None, :]
        assert excess.max() <= 0., "seed %d stack %s: %.3e beyond its bound at particle %d (n=%d)" % (
            seed, [f[0] for f in stack], excess.max(), int(excess.max(axis=0).argmax()), n)
    return widened


# seeds beyond the collected range whose prefix-sum stacks were 1.0e-12 ... 2.3e-12 from the sequential checker
# (Jacobi modify_orbits_forces and gr with 9 ... 9001 massive bodies among 4100 / 9001)
FORMER_MISSES = [4858, 6577, 6956, 7457]


@pytest.mark.parametrize("mode", [1, 0], ids=["exact", "fast"])
@pytest.mark.parametrize("seed", FORMER_MISSES)
def test_stacks_that_were_further_than_the_plain_bar_from_the_sequential_checker(ctx, seed, mode):
    """Each prefix-sum force of these stacks against its binary128 value: the device's parallel sums stay at the
    rounding level of the terms while the sequential checker itself is 1e-12 away on these layouts; then the
    whole stack with the bound derived from that.  In reference-order arithmetic the stack is walked in the
    reference's order and meets the plain bar."""
    cols, pp, stack, n, scan = random_case(seed)
    assert scan
    zc = zero_acc(cols)
    m = cols["m"]
    dev_worst = seq_worst = 0.
    for f in stack:
        if not is_prefix_sum_force(f):
            continue
        hi, lo, seq_err = exact_and_sequential(zc, pp, f)
        ctx.sequential_exact(False)         # (the prefix sums are what is measured, in either arithmetic)
        try:
            got, _ = run_device(ctx, zc, pp, [f], 1.0, mode)
        finally:
            ctx.sequential_exact(True)
        dev_err = err_vs_exact(got, hi, lo)
        # scale of the terms that enter a row: the body's own update and the reactions it collects, sum_j m_j |a_j| / m_i
        upd = np.abs(hi).max(axis=0)
        total = float((m * upd).sum())
        scale = upd + np.where(m > 0, (total - m * upd) / np.where(m > 0, m, 1.0), 0.0)
        scale = np.where(scale == 0., 1., scale)
        dev_rel, seq_rel = (dev_err / scale[None, :]).max(), (seq_err / scale[None, :]).max()
        record("fuzz-seed-%d:%s%s:n=%d:mode=%d" % (seed, f[0], f[1].get("coordinates", ""), n, mode), device_vs_terms=dev_rel, sequential_vs_terms=seq_rel)
        # the device stays at the rounding level of the terms; a component whose own expression cancels (r.v of a nearly
        # circular orbit) is no better in ANY fp64 evaluation: there, no further from exact than twice the checker
        assert (dev_err <= np.maximum(ROUNDING * scale[None, :], 2. * seq_err)).all(), (f, dev_rel)
        dev_worst, seq_worst = max(dev_worst, dev_rel), max(seq_worst, seq_rel)
    assert seq_worst >= 5. * dev_worst, (seq_worst, dev_worst)      # what moved these cases is the checker's sequential sum
    if mode == 1:
        # reference-order arithmetic walks the reference's order: every body that is not the target of a parallel
        # sum of pushes carries the checker's bits for a stack of the prefix-sum forces alone
        only = [f for f in stack if is_prefix_sum_force(f)]
        got, _ = run_device(ctx, cols, pp, only, 1.0, mode)
        assert ulp_diff(got, run_oracle(cols, pp, only)).max() == 0
    widened = run_case(ctx, seed, mode)
    assert not (mode == 1 and widened)


def random_tail_case(seed, planets_max):
    """Star (+ planets) + massless grains, random stack of forces that act on such a layout."""
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.choice([2, 3, 66, 513, 2050]))
    planets = int(rng.integers(0, min(planets_max, n - 1) + 1))
    cols, pp, _ = synthetic.disk(n, planets, seed=int(rng.integers(1, 100)))
    h = 1 + planets
    for key in ("beta", "tau_a", "tau_e", "tau_inc"):
        if n > 4 and rng.random() < 0.5:
            pp[key] = pp[key].copy()
            pp[key][rng.choice(np.arange(1, n), size=max(1, n // 5), replace=False)] = np.nan
    pp["primary"] = np.zeros(n, dtype=np.uint8)
    pp["primary"][int(rng.integers(0, h))] = 1
    k = int(rng.integers(1, 6))
    stack, seen_gr, seen_com = [], False, False
    for j in rng.permutation(sorted(rng.choice(len(FORCES), size=k, replace=False))):
        name, ep = FORCES[j]
        com = name == "modify_orbits_forces" and ep.get("coordinates", 0) in (0, 1)
        if name == "central_force" or (name == "gr" and seen_gr) or (com and seen_com):
            continue
        seen_gr |= name == "gr"
        seen_com |= com
        stack.append((name, ep))
    if not stack:
        stack = [FORCES[0]]
    return cols, pp, stack, n


@pytest.mark.parametrize("mode", [1, 0], ids=["exact", "fast"])
@pytest.mark.parametrize("seed", range(40))
def test_random_stack_through_the_force_integrators(ctx, seed, mode):
    """integrate_force (every scheme) on a random stack: reference-order arithmetic carries the
    checker's bits for the explicit schemes; the default arithmetic meets 1e-12 of the velocity change."""
    import oracle
    from reboundx_amd.device import DeviceParticles
    from reboundx_amd.program import build_ops
    cols, pp, stack, n = random_tail_case(seed, 4)
    integrator = ["euler", "rk2", "rk4", "implicit_midpoint"][seed % 4]
    ref = oracle.Sim(cols)
    oracle.integrate_force(ref, integrator, 0.01, stack, pp)
    host = dict(cols)
    for k in ("beta", "R_tides", "k1", "tau_a", "tau_e", "tau_inc"):
        host[k] = pp[k]
    dp = DeviceParticles.from_host(host)
    ctx.integrate_force(dp.view(), build_ops(stack, pp), integrator, 0.01, mode)
    v0 = np.array([cols[k] for k in ("vx", "vy", "vz")])
    vr = np.array([ref.cols[k] for k in ("vx", "vy", "vz")])
    vd = np.array([dp.cols[k].cpu().numpy() for k in ("vx", "vy", "vz")])
    assert np.isfinite(vd).all()
    if mode == 1 and integrator != "implicit_midpoint":
        assert np.array_equal(vd, vr), (seed, integrator, [f[0] for f in stack])
    scale = np.abs(vr - v0).max(axis=0)
    den = np.maximum(np.where(scale == 0., 1., scale), 4 * np.finfo(float).eps * np.abs(vr).max(axis=0) / TOL)
    assert (np.abs(vd - vr) / den[None, :]).max() <= TOL, (seed, integrator, [f[0] for f in stack])


@pytest.mark.parametrize("mode", [1, 0], ids=["exact", "fast"])
@pytest.mark.parametrize("seed", range(24))
def test_random_stack_through_wisdom_holman_steps(ctx, seed, mode):
    """rebxamd_wh_steps (star + massless grains) on a random stack against the checker's Wisdom-Holman restatement."""
    import oracle
    from reboundx_amd.device import DeviceParticles
    from reboundx_amd.program import build_ops
    cols, pp, stack, n = random_tail_case(seed, 0)
    cols["vx"][0], cols["vy"][0] = 2e-3, -1e-3
    ref = oracle.Sim(cols)
    oracle.whfast_steps(ref, 0.04, 9, stack, pp)
    want = np.array([ref.cols[k] for k in ("x", "y", "z", "vx", "vy", "vz")])
    host = dict(cols)
    for k in ("beta", "R_tides", "k1", "tau_a", "tau_e", "tau_inc"):
        host[k] = pp[k]
    dp = DeviceParticles.from_host(host, n_head=1, tail_massless=True)
    ctx.wh_steps(dp.view(), build_ops(stack, pp), 0.04, 9, mode)
    got = np.array([dp.cols[k].cpu().numpy() for k in ("x", "y", "z", "vx", "vy", "vz")])
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max(), (seed, [f[0] for f in stack])
