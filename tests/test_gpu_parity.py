"""Parity of the HIP path (through the C ABI of librebxamd.so) against the CPU checker.

Bars (fp64 work; stated by the project's north star, restated here):
  * per-particle forces: <= 1e-12 relative per acceleration component;
  * gr_full after its sweep loop: <= 1e-10 relative per component.
How "relative" is measured is defined in gpu_common.rel_err: against the component itself,
floored at FLOOR x the largest component of the same particle's update.

Two arithmetic modes are tested:
  * MODE_EXACT (reference expression order, no FMA contraction): every row that is not the
    target of a parallel reduction must carry the checker's bits (0 ulp); central_force is
    the one exception (libm pow vs. the device pow differ by a few ulp);
  * MODE_FAST (shared reciprocal square roots, FMA): the 1e-12 bar.
"""
import ctypes
import hashlib

import numpy as np
import pytest

import oracle
from gpu_common import FLOOR, TOL, err_vs_exact, record, rel_err, run_device, run_oracle, ulp_diff, unfloored_rel_err
from reboundx_amd import synthetic

pytestmark = pytest.mark.gpu

C = synthetic.C_DEFAULT
EXACT, FAST = 1, 0


@pytest.fixture(scope="module")
def ctx():
    from reboundx_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def zero_acc(cols):
    c = dict(cols)
    for k in ("ax", "ay", "az"):
        c[k] = np.zeros_like(cols["x"])
    return c


def check(ctx, cols, pp, forces, mode, G=1.0, reduced_rows=(), pow_ulps=0, tol=TOL, **kw):
    ref = run_oracle(cols, pp, forces, G)
    got, _ = run_device(ctx, cols, pp, forces, G, mode, **kw)
    assert np.isfinite(got).all()
    if mode == EXACT:
        u = ulp_diff(got, ref).max(axis=0)
        mask = np.ones(u.shape, dtype=bool)
        mask[list(reduced_rows)] = False
        assert u[mask].max(initial=0) <= pow_ulps, "rows differ from the checker in exact mode: %s" % np.nonzero(u > pow_ulps)[0][:10]
    e = rel_err(got, ref)
    record("check:" + "+".join(f[0] for f in forces) + ":n=%d:mode=%d" % (got.shape[1], mode),
           floored=e.max(initial=0.0), unfloored=unfloored_rel_err(got, ref).max(initial=0.0))
    assert e.max(initial=0.0) <= tol, "max rel err %.3e at particle %d" % (e.max(), int(e.argmax()))
    return got, ref


C2_STACK = [("radiation_forces", {"c": C}), ("gravitational_harmonics", {}), ("gr", {"c": C})]


@pytest.mark.parametrize("mode", [EXACT, FAST])
@pytest.mark.parametrize("n,n_planets", [(2, 0), (3, 0), (1000, 0), (3001, 2), (4098, 5), (70001, 1)])
def test_debris_disk_stack_head_plus_massless_tail(ctx, n, n_planets, mode):
    """radiation_forces + gravitational_harmonics + gr on star(+planets) + massless dust."""
    cols, pp, _ = synthetic.disk(n, n_planets)
    check(ctx, zero_acc(cols), pp, C2_STACK, mode)
    synthetic.add_newtonian(cols, n_head=1 + n_planets)
    check(ctx, cols, pp, C2_STACK, mode)


@pytest.mark.parametrize("mode", [EXACT, FAST])
@pytest.mark.parametrize("force", [("radiation_forces", {"c": C}), ("gravitational_harmonics", {}), ("gr", {"c": C}),
                                   ("gr", {"c": 50.0, "max_iterations": 3}), ("gr_potential", {"c": C}), ("tides_precession", {}),
                                   ("modify_orbits_forces", {}), ("modify_orbits_forces", {"coordinates": 1}),
                                   ("modify_orbits_forces", {"coordinates": 2})])
def test_each_force_head_plus_massless_tail(ctx, force, mode):
    n = 5003
    cols, pp, _ = synthetic.disk(n, 3)
    pp["primary"] = np.zeros(n, dtype=np.uint8)
    pp["primary"][0] = 1
    check(ctx, zero_acc(cols), pp, [force], mode)


@pytest.mark.parametrize("mode", [EXACT, FAST])
@pytest.mark.parametrize("gamma", [-1.0, -2.5, 1.0])
def test_central_force(ctx, gamma, mode):
    n = 2001
    cols, pp, _ = synthetic.disk(n, 2)
    pp["Acentral"] = np.full(n, np.nan)
    pp["gammacentral"] = np.full(n, np.nan)
    pp["Acentral"][0], pp["gammacentral"][0] = 1e-4, gamma
    # pow() is the one libm call on the path; device and host libm agree to a few ulp only
    check(ctx, zero_acc(cols), pp, [("central_force", {})], mode, pow_ulps=16)


C5_STACK = [("radiation_forces", {"c": C}), ("gravitational_harmonics", {}), ("gr_potential", {"c": C}),
            ("tides_precession", {}), ("modify_orbits_forces", {"coordinates": 2})]
FIXED_STACKS = [C2_STACK, C5_STACK, [("radiation_forces", {"c": C})], [("gravitational_harmonics", {})], [("gr", {"c": C})],
                [("gr_potential", {"c": C})], [("tides_precession", {})], [("modify_orbits_forces", {"coordinates": 2})],
                [("tides_precession", {}), ("modify_orbits_forces", {"coordinates": 2}), ("gr_potential", {"c": C})],
                [("tides_precession", {}), ("gr_potential", {"c": C})]]


@pytest.mark.parametrize("mode", [EXACT, FAST])
@pytest.mark.parametrize("dust_mass", [0.0, 1e-13])
@pytest.mark.parametrize("stack", FIXED_STACKS, ids=lambda s: "+".join(f[0] for f in s))
def test_kernel_instances_built_for_a_stack_agree_with_the_op_list_walk(ctx, stack, dust_mass, mode):
    """Stacks with a kernel instance of their own (massless tail and all-massive layouts) against
    the same stack run by walking the op list, and both against the checker."""
    n = 40003
    cols, pp, _ = synthetic.disk(n, 2 if dust_mass == 0.0 else 0, dust_mass=dust_mass)
    pp["primary"] = np.zeros(n, dtype=np.uint8)
    pp["primary"][0] = 1
    synthetic.add_newtonian(cols, n_head=3 if dust_mass == 0.0 else 1)
    if dust_mass != 0.0 and any(f[0] == "gr" for f in stack):
        pytest.skip("gr with every particle massive runs on the prefix-sum passes")
    ref = run_oracle(cols, pp, stack)
    try:
        ctx.fixed_stacks(False)
        walked, _ = run_device(ctx, cols, pp, stack, 1.0, mode)
    finally:
        ctx.fixed_stacks(True)
    fixed, _ = run_device(ctx, cols, pp, stack, 1.0, mode)
    nhead = 3 if dust_mass == 0.0 else 1
    if mode == EXACT:
        assert ulp_diff(fixed[:, nhead:], walked[:, nhead:]).max() == 0
    for got in (walked, fixed):
        assert rel_err(got[:, nhead:], ref[:, nhead:]).max() <= TOL
        # rows of bodies that collect pushes from the others: bar relative to the summed terms
        scale = np.maximum(np.abs(ref[:, :nhead]).max(), np.abs(ref[:, nhead:] - cols_acc(cols)[:, nhead:]).max() * max(dust_mass, 1e-300) * n)
        assert np.abs(got[:, :nhead] - ref[:, :nhead]).max() <= 1e-12 * scale


def cols_acc(cols):
    return np.stack([cols["ax"], cols["ay"], cols["az"]])


C3_STACK = [("tides_precession", {}), ("modify_orbits_forces", {"coordinates": 2}), ("gr_potential", {"c": C})]


@pytest.mark.parametrize("mode", [EXACT, FAST])
@pytest.mark.parametrize("n", [100, 20001, 65537])
def test_massive_everywhere_stack_with_back_reactions(ctx, n, mode):
    """tides + modify_orbits_forces(PARTICLE) + gr_potential where every particle has mass:
    the push on the source is a parallel reduction, so row 0 is compared with the 1e-12 bar
    relative to the size of the terms that were summed."""
    cols, pp, _ = synthetic.disk(n, 1, dust_mass=1e-15)
    pp["primary"] = np.zeros(n, dtype=np.uint8)
    pp["primary"][0] = 1
    cols = zero_acc(cols)
    ref = run_oracle(cols, pp, C3_STACK)
    got, _ = run_device(ctx, cols, pp, C3_STACK, 1.0, mode)
    u = ulp_diff(got, ref).max(axis=0)
    if mode == EXACT:
        assert u[1:].max() == 0
    assert rel_err(got[:, 1:], ref[:, 1:]).max() <= TOL
    # source row: scale = sum of |terms| ~ n * typical push; bound it from the per-particle pushes
    push_scale = np.abs(ref[:, 1:]).max() * 1e-15 * n          # m_dust/m_star * |a| summed
    assert np.abs(got[:, 0] - ref[:, 0]).max() <= 1e-12 * max(push_scale, np.abs(ref[:, 0]).max())


@pytest.mark.parametrize("mode", [EXACT, FAST])
def test_each_back_reacting_force_massive_everywhere(ctx, mode):
    n = 30011
    cols, pp, _ = synthetic.disk(n, 1, dust_mass=1e-12)
    pp["primary"] = np.zeros(n, dtype=np.uint8)
    pp["primary"][0] = 1
    pp["Acentral"] = np.full(n, np.nan); pp["gammacentral"] = np.full(n, np.nan)
    pp["Acentral"][0], pp["gammacentral"][0] = 1e-4, -1.0
    cols = zero_acc(cols)
    for force in (("gravitational_harmonics", {}), ("gr_potential", {"c": C}), ("tides_precession", {}),
                  ("central_force", {}), ("modify_orbits_forces", {"coordinates": 2}), ("radiation_forces", {"c": C})):
        ref = run_oracle(cols, pp, [force])
        got, _ = run_device(ctx, cols, pp, [force], 1.0, mode)
        if mode == EXACT and force[0] != "central_force":
            assert ulp_diff(got[:, 1:], ref[:, 1:]).max() == 0, force
        assert rel_err(got[:, 1:], ref[:, 1:]).max() <= TOL, force
        scale = max(np.abs(ref[:, 1:]).max() * 1e-12 * n, np.abs(ref[:, 0]).max())
        assert np.abs(got[:, 0] - ref[:, 0]).max() <= 1e-12 * scale, force


_EXACT_CACHE = {}
SEQ_MAX = 65536            # REBXAMD_SEQ_MAX_BODIES: up to here reference-order arithmetic walks the recurrences in the reference's order
ROUNDING = 5e-14           # |device - exact| of the prefix-sum forces relative to the body's largest component (measured: <= 2.1e-14)
HP_MAX_N = 500_000          # the binary128 evaluation is software arithmetic: ~10 s per 1e6 bodies and force


def is_prefix_sum_force(force):
    return force[0] == "gr" or (force[0] == "modify_orbits_forces" and force[1].get("coordinates", 0) in (0, 1))


def exact_and_sequential(cols, pp, force):
    """One prefix-sum force alone on `cols` (its update does not depend on the incoming
    accelerations): the update evaluated in binary128 (hi, lo), and the distance of the
    sequential fp64 checker -- the reference's order of operations -- from it, per component."""
    digest = hashlib.md5(b"".join(np.ascontiguousarray(cols[k]).tobytes() for k in ("x", "y", "z", "vx", "vy", "vz", "m"))).hexdigest()
    key = (digest, force[0], tuple(sorted(force[1].items())))
    if key not in _EXACT_CACHE:
        zc = zero_acc(cols)
        hi, lo = oracle.hp_force(oracle.Sim(zc), force, pp)
        seq = run_oracle(zc, pp, [force])
        _EXACT_CACHE[key] = (hi, lo, err_vs_exact(seq, hi, lo))
        if len(_EXACT_CACHE) > 12:
            _EXACT_CACHE.pop(next(iter(_EXACT_CACHE)))
    return _EXACT_CACHE[key]


def scan_check(ctx, cols, pp, forces, mode, scans=False, **kw):
    """Layouts with massive bodies anywhere.  Reference-order arithmetic on at most SEQ_MAX bodies
    (and `scans` not forced): the device walks the reference's recurrences in the reference's order,
    so a stack made of gr and Jacobi/barycentric modify_orbits_forces alone must carry the checker's
    bits on every body.  Otherwise those two forces are
    evaluated with parallel prefix/suffix sums where the reference accumulates sequentially, so
    the device and the fp64 checker are two different roundings of the same real-number result.
    The bound on their difference is derived, not chosen: with E the binary128 value of the update
    (oracle/rebx_oracle_hp.c),
        |device - checker| <= |device - E| + |checker - E|,
    the first term held to 1e-12 of the component, but no tighter than ROUNDING (5e-14) of the body's
    largest component -- the rounding of the terms of these formulas, which no fp64 evaluation, the
    reference's included, gets below in a component that nearly cancels -- and the second MEASURED
    per component: it is the reference's own accumulation error, which grows with the number of
    bodies (test_prefix_sum_forces_against_the_binary128_value: at 6000 bodies the sequential
    checker is 1e-12 of a body's largest component from E, the device 2e-14).  Beyond HP_MAX_N
    bodies the binary128 evaluation is too slow; there the second term is bounded by 1e-12 of the
    body's largest component or of the summed magnitudes of the reaction terms that enter the row
    (sum_k m_k/M |a_k|)."""
    ref = run_oracle(cols, pp, forces)
    ctx.sequential_exact(not scans)
    try:
        got, _ = run_device(ctx, cols, pp, forces, 1.0, mode, **kw)
    finally:
        ctx.sequential_exact(True)
    assert np.isfinite(got).all()
    massive = np.nonzero(cols["m"])[0]
    chain = kw.get("n_head") or (int(massive[-1]) + 1 if len(massive) else 1)      # (massless bodies behind the last massive one ride free)
    if kw.get("tail_massless") is False:
        chain = got.shape[1]
    if mode == EXACT and not scans and chain <= SEQ_MAX and all(is_prefix_sum_force(f) for f in forces):
        u = ulp_diff(got, ref).max(axis=0)
        assert u.max() == 0, "reference-order arithmetic: %d bodies differ from the checker, first %d" % (int((u > 0).sum()), int(np.nonzero(u)[0][0]))
        return got, ref
    upd = ref - np.array([cols["ax"], cols["ay"], cols["az"]])
    big = np.maximum(np.abs(ref).max(axis=0), np.abs(upd).max(axis=0))
    allowed = np.maximum(TOL * np.abs(ref), ROUNDING * big[None, :])
    n = got.shape[1]
    if n <= HP_MAX_N:
        for f in forces:
            if is_prefix_sum_force(f):
                allowed = allowed + exact_and_sequential(cols, pp, f)[2]
    else:
        m = cols["m"]
        react = float((m[1:] / m.sum() * np.abs(upd[:, 1:]).max(axis=0)).sum())
        allowed = allowed + TOL * np.maximum(big, react)[None, :]
    excess = np.abs(got - ref) - allowed
    record("scan:" + "+".join(f[0] for f in forces) + ":n=%d:mode=%d" % (n, mode),
           floored=rel_err(got, ref).max(initial=0.0), unfloored=unfloored_rel_err(got, ref).max(initial=0.0))
    assert excess.max() <= 0., "component off by %.3e beyond its bound %.3e at particle %d" % (
        excess.max(), allowed.flat[excess.argmax()], int(excess.max(axis=0).argmax()))
    return got, ref


SCAN_FORCES = [("gr", {"c": C}), ("gr", {"c": 300.0, "max_iterations": 4}), ("modify_orbits_forces", {}),
               ("modify_orbits_forces", {"coordinates": 1})]


@pytest.mark.parametrize("mode", [EXACT, FAST])
@pytest.mark.parametrize("force", SCAN_FORCES)
@pytest.mark.parametrize("n", [70, 257, 6000])
def test_gr_and_com_drag_with_every_particle_massive(ctx, n, force, mode):
    """Reference-order arithmetic: the checker's bits (asserted in scan_check); the same arithmetic with the
    prefix sums forced, and the default arithmetic: the derived bound."""
    cols, pp, _ = synthetic.disk(n, 2, dust_mass=1e-9)
    scan_check(ctx, zero_acc(cols), pp, [force], mode)
    if mode == EXACT:
        scan_check(ctx, zero_acc(cols), pp, [force], mode, scans=True)


@pytest.mark.parametrize("force", SCAN_FORCES)
def test_reference_order_route_with_incoming_accelerations_and_odd_sizes(ctx, force):
    """The route that walks the reference's recurrences in order, around the 256-body tiles of its
    chains (255, 256, 257, 513 bodies), with accelerations already in the columns, and on two bodies."""
    for n in (2, 3, 255, 256, 257, 513, 1300):
        cols, pp, _ = synthetic.disk(n, min(2, n - 1), dust_mass=1e-7)
        synthetic.add_newtonian(cols, n_head=n)
        got, ref = scan_check(ctx, cols, pp, [force], EXACT, n_head=n, tail_massless=False)
        assert ulp_diff(got, ref).max() == 0


def test_reference_order_route_stops_at_its_size_limit(ctx):
    """With the last massive body above index SEQ_MAX - 1 reference-order arithmetic uses the prefix sums (derived
    bound); at the limit it still carries the checker's bits."""
    force = ("modify_orbits_forces", {})
    for n, bits in ((SEQ_MAX, True), (SEQ_MAX + 1, False)):
        cols, pp, _ = synthetic.disk(n, 2)
        rng = np.random.default_rng(n)
        cols["m"][rng.choice(np.arange(3, n), size=40, replace=False)] = 1e-6
        cols["m"][n - 1] = 1e-6                 # (the chains reach up to the last massive body)
        got, ref = scan_check(ctx, zero_acc(cols), pp, [force], EXACT)
        assert (ulp_diff(got, ref).max() == 0) == bits


@pytest.mark.parametrize("force", SCAN_FORCES)
def test_reference_order_route_with_massless_grains_behind_the_last_massive_body(ctx, force):
    """Massive bodies scattered over a prefix (beyond the 64-body head of the fused path), then massless grains,
    more of them than the chains' size limit: the chains cover the prefix, the grains are measured from the
    final running sums, and every body carries the checker's bits.  A one-grain and a two-grain tail check
    the count of re-roundings of the centre of mass; accelerations are already in the columns."""
    for n, prefix, massive in ((SEQ_MAX + 3000, 700, 90), (1301, 1300, 1300), (1302, 1300, 200), (5000, 257, 257)):
        cols, pp, _ = synthetic.disk(n, 2)
        rng = np.random.default_rng(n)
        cols["m"][rng.choice(np.arange(3, prefix - 1), size=massive - 4, replace=False)] = 10.0 ** rng.uniform(-9, -5, massive - 4)
        cols["m"][prefix - 1] = 3e-7
        cols["m"][prefix:] = 0.
        synthetic.add_newtonian(cols, n_head=prefix)
        got, ref = scan_check(ctx, cols, pp, [force], EXACT)
        assert ulp_diff(got, ref).max() == 0


def exact_value_check(ctx, cols, pp, force, mode, label, **kw):
    """|device - E| against the bar and against |sequential checker - E|, E = binary128 value."""
    cols = zero_acc(cols)
    hi, lo, seq_err = exact_and_sequential(cols, pp, force)
    ctx.sequential_exact(False)        # (the prefix-sum passes are what is measured here, in either arithmetic)
    try:
        got, _ = run_device(ctx, cols, pp, [force], 1.0, mode, **kw)
    finally:
        ctx.sequential_exact(True)
    dev_err = err_vs_exact(got, hi, lo)
    mag = np.abs(hi)
    body = mag.max(axis=0)
    body = np.where(body == 0., 1., body)
    with np.errstate(divide="ignore", invalid="ignore"):
        dev_unfloored = np.where(mag > 0, dev_err / mag, 0.).max()
        seq_unfloored = np.where(mag > 0, seq_err / mag, 0.).max()
    dev_floored = (dev_err / np.maximum(mag, FLOOR * body[None, :])).max()
    seq_floored = (seq_err / np.maximum(mag, FLOOR * body[None, :])).max()
    dev_vec, seq_vec = (dev_err / body[None, :]).max(), (seq_err / body[None, :]).max()
    record("exact:%s:%s:n=%d:mode=%d" % (label, force[0] + str(force[1].get("coordinates", "")), got.shape[1], mode),
           device_floored=dev_floored, device_unfloored=dev_unfloored, device_vs_largest_component=dev_vec,
           sequential_floored=seq_floored, sequential_unfloored=seq_unfloored, sequential_vs_largest_component=seq_vec)
    # (1) relative to the body's largest component the device stays at the rounding level of the terms, whatever
    #     the length of the sums; hence every component of at least 5 % of its body's largest one is within
    #     1e-12 of the exact value (the bar of every other force)
    assert dev_vec <= ROUNDING, dev_vec
    well = mag >= 0.05 * body[None, :]
    assert (dev_err[well] <= TOL * mag[well]).all()
    # (2) components that nearly cancel lose digits in ANY fp64 evaluation: there, with the floor of the other
    #     tests (1e-3 of the body's largest component) and with no floor at all, the device is held to the bar or
    #     to twice the distance of the reference's own sequential evaluation from the exact value
    assert dev_floored <= max(TOL, 2. * seq_floored), (dev_floored, seq_floored)
    assert dev_unfloored <= max(TOL, 2. * seq_unfloored), (dev_unfloored, seq_unfloored)
    return dev_vec, seq_vec


@pytest.mark.parametrize("mode", [EXACT, FAST])
@pytest.mark.parametrize("force", SCAN_FORCES)
@pytest.mark.parametrize("layout", ["all-massive-70", "all-massive-257", "all-massive-6000", "planets-scattered-9001", "long-prefix-4000",
                                    "all-massive-30000"])
def test_prefix_sum_forces_against_the_binary128_value(ctx, layout, force, mode):
    """How far the parallel prefix/suffix-sum evaluation and the reference's sequential one are from
    the exact value of the same formulas.  The sequential sums accumulate rounding with the number of
    bodies (at 6000 bodies of 1e-9 solar masses the fp64 checker is ~1e-12 of a body's largest
    component away from the exact value, at 300 000 bodies 5e-11); the device's tree-ordered sums do not."""
    kw = {}
    if layout.startswith("all-massive"):
        n = int(layout.split("-")[-1])
        if force[0] == "gr" and n > 6000:
            pytest.skip("gr's Newtonian term costs n x massive pair terms in software binary128")
        cols, pp, _ = synthetic.disk(n, 2, dust_mass=1e-9)
    elif layout.startswith("planets"):
        cols, pp, _ = synthetic.disk(9001, 6)
        cols, pp = scatter_planets(cols, pp)
    else:
        cols, pp, _ = synthetic.disk(4000, 99)
        kw = dict(n_head=100, tail_massless=True)
    dev_vec, seq_vec = exact_value_check(ctx, cols, pp, force, mode, layout, **kw)
    if layout in ("all-massive-6000", "all-massive-30000") and force[0] == "modify_orbits_forces":
        assert seq_vec > 10 * dev_vec          # the long sequential sums are what moves the checker, not the device


def scatter_planets(cols, pp, seed=5):
    """Moves the massive bodies behind the star to random slots (the star stays at index 0)."""
    n = len(cols["x"])
    perm = np.concatenate([[0], 1 + np.random.default_rng(seed).permutation(n - 1)])
    c = {k: np.ascontiguousarray(v[perm]) for k, v in cols.items()}
    q = {k: (np.ascontiguousarray(v[perm]) if isinstance(v, np.ndarray) and v.shape[:1] == (n,) else v) for k, v in pp.items()}
    return c, q


@pytest.mark.parametrize("mode", [EXACT, FAST])
@pytest.mark.parametrize("force", SCAN_FORCES)
def test_gr_and_com_drag_with_planets_scattered_among_massless_dust(ctx, force, mode):
    n = 9001
    cols, pp, _ = synthetic.disk(n, 6)
    cols, pp = scatter_planets(cols, pp)
    assert (cols["m"][7:] != 0).any()
    cols = zero_acc(cols)
    scan_check(ctx, cols, pp, [force], mode)
    synthetic.add_newtonian(cols, n_head=n)
    scan_check(ctx, cols, pp, [force], mode)


@pytest.mark.parametrize("mode", [EXACT, FAST])
def test_more_massive_bodies_than_the_head_kernel_holds(ctx, mode):
    """100 massive bodies followed by massless dust: the prefix is longer than 64."""
    n = 4000
    cols, pp, _ = synthetic.disk(n, 99)
    for force in SCAN_FORCES:
        scan_check(ctx, zero_acc(cols), pp, [force], mode, n_head=100, tail_massless=True)


@pytest.mark.parametrize("mode", [EXACT, FAST])
def test_full_stack_with_planets_scattered(ctx, mode):
    """Streaming forces before, between and after the two scan-based ones keep their order."""
    n = 5000
    cols, pp, _ = synthetic.disk(n, 4, dust_mass=1e-11)
    cols, pp = scatter_planets(cols, pp, seed=9)
    pp["primary"] = np.zeros(n, dtype=np.uint8)
    pp["primary"][0] = 1
    stack = [("radiation_forces", {"c": C}), ("gr", {"c": C}), ("gravitational_harmonics", {}),
             ("modify_orbits_forces", {"coordinates": 1}), ("tides_precession", {}), ("modify_orbits_forces", {})]
    scan_check(ctx, zero_acc(cols), pp, stack, mode)


@pytest.mark.parametrize("layout", ["all-massive", "scattered-with-massless-tail", "too-long"])
def test_sources_collect_their_pushes_in_the_order_of_the_reference(ctx, layout):
    """Reference-order arithmetic, massive bodies beyond the 64-body head: the sources of gravitational_harmonics (two of
    them), gr_potential and tides_precession and the reference particle of modify_orbits_forces collect the pushes of
    every other body one by one -- ascending for the sourced forces, descending for modify_orbits_forces
    (src/rebxtools.c:71) -- between their own updates as particles of the other forces, so every body, the sources
    included, carries the checker's bits; the full stack with gr and Jacobi drag in the middle as well.  With the last
    massive body above the size limit of the chains the pushes are summed per workgroup: the sources are then a few ulp
    from the checker, the rest of the bodies still bitwise."""
    if layout == "all-massive":
        n = 6000
        cols, pp, _ = synthetic.disk(n, 3, dust_mass=1e-10)
    else:
        n = SEQ_MAX + 2500
        cols, pp, _ = synthetic.disk(n, 3)
        rng = np.random.default_rng(5)
        last = 900 if layout == "scattered-with-massless-tail" else n - 7
        cols["m"][rng.choice(np.arange(70, last), size=60, replace=False)] = 10.0 ** rng.uniform(-9, -5, 60)
        cols["m"][last] = 1e-7
    pp = dict(pp)
    for key, val in (("J2", 2e-3), ("J4", -1e-4), ("R_eq", 2e-3)):
        pp[key] = pp[key].copy(); pp[key][1] = val
    pp["primary"] = np.zeros(n, dtype=np.uint8)
    pp["primary"][2] = 1
    synthetic.add_newtonian(cols, n_head=4)
    pushing = [("gravitational_harmonics", {}), ("modify_orbits_forces", {"coordinates": 2}), ("gr_potential", {"c": C}), ("tides_precession", {}),
               ("radiation_forces", {"c": C})]
    stacks = [pushing]
    if layout != "too-long":
        stacks.append(pushing[:2] + [("gr", {"c": C}), ("modify_orbits_forces", {})] + pushing[2:])
    for stack in stacks:
        ref = run_oracle(cols, pp, stack)
        got, _ = run_device(ctx, cols, pp, stack, 1.0, EXACT)
        u = ulp_diff(got, ref).max(axis=0)
        if layout == "too-long":
            assert u[3:].max() == 0 and 0 < u[:3].max() < 4096
        else:
            assert u.max() == 0, np.nonzero(u)[0][:8]


RIDING_STACKS = {
    "tides+jacobi+gr_potential": [("tides_precession", {}), ("modify_orbits_forces", {}), ("gr_potential", {"c": C})],
    "harmonics+barycentric": [("gravitational_harmonics", {}), ("modify_orbits_forces", {"coordinates": 1})],
    "gr+gr_potential": [("gr", {"c": C}), ("gr_potential", {"c": C})],
    "tides+gr": [("tides_precession", {}), ("gr", {"c": 300.0, "max_iterations": 4})],
    "gr+harmonics+jacobi+radiation+tides": [("gr", {"c": C}), ("gravitational_harmonics", {}), ("modify_orbits_forces", {}),
                                            ("radiation_forces", {"c": C}), ("tides_precession", {})],
}


@pytest.mark.parametrize("fixed", [True, False])
@pytest.mark.parametrize("stack", sorted(RIDING_STACKS))
@pytest.mark.parametrize("n", [510, 514, 700, 2050, 8194, 20_001])        # (around the 128-, 512- and 2048-body marks of the passes; even and odd)
def test_streaming_pass_that_also_applies_a_prefix_sum_force(ctx, n, stack, fixed):
    """Default arithmetic, massive bodies anywhere: the pass of the streaming forces also adds the
    last prefix-sum force (its force per body minus the reactions of the bodies behind), so the
    accelerations are read and written once for both.  Checked against the checker, and against
    the same stack with the force applied by a pass of its own."""
    cols, pp, _ = synthetic.disk(n, 3, dust_mass=1e-11)
    cols = zero_acc(cols)
    synthetic.add_newtonian(cols, n_head=4)
    forces = RIDING_STACKS[stack]
    ctx.fixed_stacks(fixed)
    try:
        riding, _ = scan_check(ctx, cols, pp, forces, FAST)
        ctx.carried_steps(False)
        apart, _ = scan_check(ctx, cols, pp, forces, FAST)
    finally:
        ctx.carried_steps(True)
        ctx.fixed_stacks(True)
    scale = np.abs(apart).max(axis=0)
    assert (np.abs(riding - apart) <= 1e-13 * np.where(scale == 0., 1., scale)).all()


# parameters of the reference's own tests (reboundx/test/test_machine_independent.py:39-42,135-136,
# 159-160,183-184,207-209) on the reference's own three-body system (:25-31)
TWOPLANETS_PP = {
    "beta": np.array([np.nan, .5, .3]),
    "J2": np.array([1e-3, np.nan, np.nan]), "J4": np.array([1e-3, np.nan, np.nan]), "R_eq": np.array([1e-3, np.nan, np.nan]),
    "R_tides": np.array([1e-3, np.nan, np.nan]), "k1": np.array([.4, np.nan, np.nan]),
    "tau_a": np.array([np.nan, -1e4, np.nan]), "tau_e": np.array([np.nan, -1e3, -1e3]), "tau_inc": np.array([np.nan, np.nan, -1e3]),
    "Acentral": np.array([1e-3, np.nan, np.nan]), "gammacentral": np.array([-1., np.nan, np.nan]),
}


@pytest.mark.parametrize("mode", [EXACT, FAST])
@pytest.mark.parametrize("force", [("radiation_forces", {"c": 1e4}), ("gravitational_harmonics", {}), ("gr", {"c": 1e4}),
                                   ("gr_potential", {"c": 1e4}), ("tides_precession", {}), ("central_force", {}),
                                   ("modify_orbits_forces", {}), ("modify_orbits_forces", {"coordinates": 1})])
def test_reference_test_system_each_force(ctx, force, mode):
    cols = synthetic.twoplanets()
    s = oracle.Sim(cols)
    oracle.newtonian_gravity(s)
    cols["ax"], cols["ay"], cols["az"] = [a.copy() for a in s.acc]
    check(ctx, cols, TWOPLANETS_PP, [force], mode, pow_ulps=16 if force[0] == "central_force" else 0)
    check(ctx, zero_acc(cols), TWOPLANETS_PP, [force], mode, pow_ulps=16 if force[0] == "central_force" else 0)


@pytest.mark.parametrize("mode", [EXACT, FAST])
def test_all_seven_streaming_forces_stacked_in_reverse_add_order(ctx, mode):
    """The reference runs forces in the reverse of the order they were added
    (src/linkedlist.c:33-36); here the execution order is given explicitly."""
    n = 8191
    cols, pp, _ = synthetic.disk(n, 4)
    pp["primary"] = np.zeros(n, dtype=np.uint8); pp["primary"][0] = 1
    stack = [("modify_orbits_forces", {"coordinates": 0}), ("tides_precession", {}), ("gr", {"c": C}),
             ("gr_potential", {"c": C}), ("gravitational_harmonics", {}), ("radiation_forces", {"c": C}),
             ("modify_orbits_forces", {"coordinates": 2})]
    synthetic.add_newtonian(cols, n_head=5)
    check(ctx, cols, pp, stack, mode)


@pytest.mark.parametrize("mode", [EXACT, FAST])
def test_several_sources(ctx, mode):
    """Two radiation sources, two oblate bodies, two tidal primaries: one pass per source, in
    index order, J2 passes before J4 passes (gravitational_harmonics.c:137-140)."""
    n = 4000
    cols, pp, _ = synthetic.disk(n, 3)
    pp["radiation_source"] = np.zeros(n, dtype=np.uint8); pp["radiation_source"][[0, 2]] = 1
    pp["tides_primary"] = np.zeros(n, dtype=np.uint8); pp["tides_primary"][[0, 1]] = 1
    pp["J2"][2], pp["J4"][2], pp["R_eq"][2] = 2e-2, 1e-3, 5e-3
    pp["J2"][3] = 0.5          # J2 without R_eq: ignored (gravitational_harmonics.c:90-92)
    stack = [("radiation_forces", {"c": C}), ("gravitational_harmonics", {}), ("tides_precession", {})]
    check(ctx, zero_acc(cols), pp, stack, mode)


def test_particles_without_beta_are_untouched(ctx):
    n = 3000
    cols, pp, _ = synthetic.disk(n, 0)
    pp["beta"][100:200] = np.nan
    synthetic.add_newtonian(cols, n_head=1)
    got, ref = check(ctx, cols, pp, [("radiation_forces", {"c": C})], EXACT)
    for k, name in enumerate(("ax", "ay", "az")):
        assert np.array_equal(got[k, 100:200], cols[name][100:200])
        assert np.array_equal(got[k, 0:1], cols[name][0:1])        # the source itself


@pytest.mark.parametrize("mode", [EXACT, FAST])
def test_nans_that_host_arithmetic_produces_mean_not_set(ctx, mode):
    """Parameter columns computed on the host carry NaNs of other bit patterns than numpy.nan: the negated
    marker (-tau with unset entries), 0/0 and inf - inf (sign bit set on x86), NaNs with a payload.  All of
    them mean "not set"; only the one pattern reserved for a NaN-valued parameter is a value."""
    n = 3000
    cols, pp, _ = synthetic.disk(n, 2, dust_mass=1e-12)
    pp["primary"] = np.zeros(n, dtype=np.uint8); pp["primary"][0] = 1
    rng = np.random.default_rng(4)
    for key in ("beta", "tau_a", "tau_e", "tau_inc", "R_tides", "k1"):
        pp[key] = pp[key].copy()
        pp[key][rng.choice(np.arange(1, n), size=n // 3, replace=False)] = np.nan
    stack = C5_STACK + [("modify_orbits_forces", {})]
    want = run_oracle(zero_acc(cols), pp, stack)
    odd = dict(pp)
    with np.errstate(invalid="ignore", divide="ignore"):
        zero_over_zero = np.zeros(1) / np.zeros(1)
        odd["tau_a"] = -np.where(np.isnan(pp["tau_a"]), np.nan, -pp["tau_a"])
        odd["tau_e"] = np.where(np.isnan(pp["tau_e"]), zero_over_zero[0], pp["tau_e"])
        odd["beta"] = np.where(np.isnan(pp["beta"]), np.inf - np.inf, pp["beta"])
        payload = np.array([0x7FF8000000000123], dtype=np.uint64).view(np.float64)[0]
        odd["tau_inc"] = np.where(np.isnan(pp["tau_inc"]), payload, pp["tau_inc"])
        signalling = np.array([0xFFF0000000000001], dtype=np.uint64).view(np.float64)[0]
        odd["k1"] = np.where(np.isnan(pp["k1"]), signalling, pp["k1"])
    patterns = {int(v) for k in ("tau_a", "tau_e", "beta", "tau_inc", "k1") for v in np.unique(odd[k][np.isnan(odd[k])].view(np.uint64))}
    assert len(patterns) >= 3 and any(p >> 63 for p in patterns)   # several patterns, a sign-flipped NaN among them
    assert np.array_equal(run_oracle(zero_acc(cols), odd, stack), want)     # the checker reads them the same way
    got, _ = run_device(ctx, zero_acc(cols), odd, stack, 1.0, mode)
    same, _ = run_device(ctx, zero_acc(cols), pp, stack, 1.0, mode)
    assert np.isfinite(got).all()
    assert np.array_equal(got, same)


def test_unaligned_columns_take_the_scalar_path(ctx):
    """Views that start 8 bytes into an allocation cannot use 16-byte accesses."""
    import torch
    from reboundx_amd.device import DeviceParticles, apply_forces
    n = 3001
    cols, pp, _ = synthetic.disk(n, 2)
    host = dict(zero_acc(cols)); host["beta"] = pp["beta"]
    ref = run_oracle(zero_acc(cols), pp, C2_STACK)
    dev = {}
    for k, v in host.items():
        if isinstance(v, np.ndarray) and v.dtype == np.float64 and k in ("x", "y", "z", "vx", "vy", "vz", "m", "ax", "ay", "az", "beta"):
            buf = torch.empty(n + 1, dtype=torch.float64, device="cuda:0")
            buf[1:] = torch.from_numpy(v)
            dev[k] = buf[1:]
    dp = DeviceParticles(dev, G=1.0, n_head=3, tail_massless=True)
    assert dp.cols["x"].data_ptr() % 16 == 8
    apply_forces(ctx, dp, C2_STACK, pp, EXACT)
    got = np.array(dp.acc_host())
    assert ulp_diff(got, ref).max() == 0


def test_classify_and_reset(ctx):
    import torch
    from reboundx_amd.device import DeviceParticles
    m = np.zeros(100000); m[0] = 1.0; m[37] = 1e-30
    t = torch.from_numpy(m).cuda()
    assert ctx.classify(t) == 38
    assert ctx.classify(torch.zeros(10, dtype=torch.float64, device="cuda")) == 1
    cols, pp, _ = synthetic.disk(1000, 0)
    synthetic.add_newtonian(cols, n_head=1)
    dp = DeviceParticles.from_host(cols)
    ctx.reset_accelerations(dp.view())
    assert all((a == 0).all() for a in dp.acc_host())


def test_gr_nonconvergence_raises_the_reference_warning_bit(ctx):
    cols = synthetic.twoplanets()
    ctx.fetch_warnings()
    run_device(ctx, zero_acc(cols), {}, [("gr", {"c": 1.2})], 1.0, EXACT)     # absurdly slow light: iteration cannot settle
    assert ctx.fetch_warnings() & 1
    run_device(ctx, zero_acc(cols), {}, [("gr", {"c": 1e4})], 1.0, EXACT)
    assert ctx.fetch_warnings() == 0


# --------------------------------------------------------------------------- gr_full

def cluster_with_gravity(n):
    cl = synthetic.cluster(n)
    s = oracle.Sim(cl)
    oracle.newtonian_gravity(s)
    cl["ax"], cl["ay"], cl["az"] = [a.copy() for a in s.acc]
    return cl


@pytest.mark.parametrize("mode", [EXACT, FAST])
@pytest.mark.parametrize("ignore10", [0, 1])
@pytest.mark.parametrize("n", [1, 2, 3, 16, 128, 257, 700])
def test_gr_full_against_checker(ctx, n, ignore10, mode):
    from reboundx_amd.device import DeviceParticles
    cl = cluster_with_gravity(n)
    ref = oracle.Sim(cl)
    sweeps_ref = oracle.gr_full(ref, 1e2, ignore10)
    dp = DeviceParticles.from_host(cl)
    sweeps = ctx.grfull(dp.view(), 1e2, ignore10, mode)
    got = np.array(dp.acc_host())
    a0 = np.array([cl["ax"], cl["ay"], cl["az"]])
    d_ref, d_got = np.array(ref.acc) - a0, got - a0
    if mode == EXACT:
        assert sweeps == sweeps_ref
        assert ulp_diff(got, np.array(ref.acc)).max() == 0
    else:
        # the stopping rule asks for bitwise-repeating iterates, so a last-bit difference may
        # cost or save one sweep; the iterates themselves agree far below the bar by then
        assert abs(sweeps - sweeps_ref) <= 1
        assert rel_err(d_got, d_ref).max(initial=0.0) <= 1e-10
        assert rel_err(got, np.array(ref.acc)).max(initial=0.0) <= 1e-12


@pytest.mark.parametrize("mode", [EXACT, FAST])
def test_gr_full_in_one_launch_equals_the_queued_passes_bit_for_bit(ctx, mode):
    """Up to 256 bodies rebxamd_grfull runs the whole evaluation in one launch of one workgroup: the same per-row
    functions in the same order as the queued passes, so the same bits and the same sweep count -- for 1 ... 256
    bodies, both settings of gravity_ignore_terms, a speed of light that takes all ten sweeps (with the reference's
    warning) and one that converges early.  257 bodies take the queued passes either way."""
    from reboundx_amd.device import DeviceParticles
    for n in (1, 2, 3, 9, 64, 65, 200, 256, 257):
        cl = cluster_with_gravity(n)
        for c, ignore10 in ((1e2, 0), (1e4, 1), (3.0, 0)):
            results = []
            for one_launch in (True, False):
                ctx.grfull_one_launch(one_launch)
                try:
                    dp = DeviceParticles.from_host(cl)
                    sweeps = ctx.grfull(dp.view(), c, ignore10, mode)
                    results.append((sweeps, np.array(dp.acc_host()), ctx.fetch_warnings()))
                finally:
                    ctx.grfull_one_launch(True)
            assert results[0][0] == results[1][0], (n, c)
            assert results[0][2] == results[1][2], (n, c)
            assert np.array_equal(results[0][1].view(np.uint64), results[1][1].view(np.uint64)), (n, c)


def test_gr_full_reference_test_system(ctx):
    """test_machine_independent.py:52-73: the three-body system, c = 1e4, after WHFast's
    Jacobi gravity (gravity_ignore_terms = 1 adds the 0-1 pair back, gr_full.c:92-102)."""
    from reboundx_amd.device import DeviceParticles
    cols = synthetic.twoplanets()
    s = oracle.Sim(cols); oracle.newtonian_gravity(s)
    cols["ax"], cols["ay"], cols["az"] = [a.copy() for a in s.acc]
    for ig in (0, 1):
        ref = oracle.Sim(cols); oracle.gr_full(ref, 1e4, ig)
        dp = DeviceParticles.from_host(cols)
        ctx.grfull(dp.view(), 1e4, ig, EXACT)
        assert ulp_diff(np.array(dp.acc_host()), np.array(ref.acc)).max() == 0


@pytest.mark.parametrize("mode", [EXACT, FAST])
def test_gr_full_row_ranges_compose(ctx, mode):
    """The three phases on two row shards + concatenation == the phases on all rows
    (what the multi-GPU driver relies on)."""
    import torch
    from reboundx_amd.device import DeviceParticles
    n, c = 1000, 1e2
    cl = cluster_with_gravity(n)
    dp = DeviceParticles.from_host(cl)
    v = dp.view()
    z = lambda: torch.zeros(n, dtype=torch.float64, device="cuda")
    an = [dp.cols[k].clone() for k in ("ax", "ay", "az")]
    ao = [0.01 * a for a in an]
    res = {}
    for label, cuts in (("whole", [0, n]), ("split", [0, 389, n])):
        pot, ac, out = z(), [z(), z(), z()], [z(), z(), z()]
        flag = torch.zeros(1, dtype=torch.int32, device="cuda")
        for r0, r1 in zip(cuts[:-1], cuts[1:]):
            ctx.grfull_pot(v, c, r0, r1, pot, mode)
        for r0, r1 in zip(cuts[:-1], cuts[1:]):
            ctx.grfull_const(v, c, r0, r1, pot, ac, mode)
        for r0, r1 in zip(cuts[:-1], cuts[1:]):
            ctx.grfull_sweep(v, c, r0, r1, an, ao, ac, out, flag, mode)
        res[label] = [t.cpu().numpy() for t in [pot] + ac + out] + [int(flag.item())]
    for a, b in zip(res["whole"][:-1], res["split"][:-1]):
        if mode == EXACT:
            assert np.array_equal(a, b)
        else:
            assert np.abs(a - b).max() <= 1e-13 * np.abs(a).max()
    assert res["whole"][-1] == res["split"][-1] == 1


# --------------------------------------------------------------------------- full size

@pytest.mark.parametrize("mode", [FAST, EXACT])
def test_ten_million_particle_disk_matches_checker_on_a_sample(ctx, mode):
    """The headline configuration (star + 1e7 massless grains, radiation + J2/J4 + gr).
    Behind the head every grain's update depends only on the head and on itself, so the
    checker run on (head + a random sample of grains) must reproduce the sampled rows of the
    full-size device result; plus: the run is deterministic (two launches, same bits)."""
    from reboundx_amd.device import DeviceParticles, apply_forces
    n = 10_000_000
    cols, pp, _ = synthetic.disk(n, 0)
    synthetic.add_newtonian(cols, n_head=1)
    host = dict(cols); host["beta"] = pp["beta"]
    outs = []
    for _ in range(2):
        dp = DeviceParticles.from_host(host, n_head=1, tail_massless=True)
        apply_forces(ctx, dp, C2_STACK, {k: pp[k][:1] for k in ("J2", "J4", "R_eq")} | {"beta": pp["beta"][:1]}, mode)
        outs.append(np.array(dp.acc_host()))
        del dp
    assert np.array_equal(outs[0], outs[1])
    rng = np.random.default_rng(7)
    pick = np.concatenate([[0], np.sort(rng.choice(np.arange(1, n), size=20000, replace=False)), [n - 1]])
    sub = {k: v[pick].copy() for k, v in cols.items()}
    subpp = {k: v[pick].copy() for k, v in pp.items()}
    ref = run_oracle(sub, subpp, C2_STACK)
    got = outs[0][:, pick]
    if mode == EXACT:
        assert ulp_diff(got, ref).max() == 0
    assert rel_err(got, ref).max() <= TOL


@pytest.mark.parametrize("mode", [FAST, EXACT])
def test_prefix_sum_paths_over_several_rounds_of_the_totals_scan(ctx, mode):
    """2.4 million bodies, six planets at random slots among massless dust.  The scan of the
    per-workgroup totals (one row per 2048 bodies, 512 rows per round) runs two whole rounds and
    a partial one, the scan of the massive-body flags (one row per 256 bodies) eighteen; every
    planet's mass has to reach all bodies behind it across those rounds, and its reaction all
    bodies in front.  The checker's cost stays O(n x massive) on this layout."""
    n = 2_400_001
    cols, pp, _ = synthetic.disk(n, 6)
    cols, pp = scatter_planets(cols, pp)
    cols = zero_acc(cols)
    for force in (SCAN_FORCES[0], SCAN_FORCES[2], SCAN_FORCES[3]):
        scan_check(ctx, cols, pp, [force], mode)


def test_ten_million_massive_bodies_keep_the_total_momentum(ctx):
    """Full size, every body massive (the checker would need O(n^2) here).  Size-independent
    property: the reactions of these forces cancel the momentum they hand out -- with the
    inclusive Jacobi reaction m_i a_i - m_i/(M_<i + m_i) a_i (M_<i + m_i) = 0, likewise about the
    barycentre, and gr's Jacobi-to-inertial transform leaves the centre of mass alone -- so
    sum_i m_i da_i vanishes to rounding (bar: 1e-10 of sum_i m_i |da_i|, i.e. rounding of 1e7
    terms and of the prefix sums behind them).  Plus: two launches give the same bits."""
    import torch
    from reboundx_amd.device import DeviceParticles, apply_forces
    n = 10_000_000
    cols, pp, _ = synthetic.disk(n, 4, dust_mass=1e-12)
    host = dict(zero_acc(cols))
    for k in ("tau_a", "tau_e", "tau_inc"):
        host[k] = pp[k]
    for force in (("modify_orbits_forces", {}), ("modify_orbits_forces", {"coordinates": 1}), ("gr", {"c": C})):
        if force[0] == "gr":
            # gr's Newtonian term pairs every body with every massive one: 1e14 pairs at this size (100 s); a fifth of the bodies keep it to 4 s
            n = 2_000_000
            host = {k: v[:n] for k, v in host.items()}
        m = torch.from_numpy(np.ascontiguousarray(host["m"])).to("cuda:0")
        outs = []
        for _ in range(2):
            dp = DeviceParticles.from_host(host, n_head=n, tail_massless=False)
            apply_forces(ctx, dp, [force], {}, FAST)
            outs.append(torch.stack([dp.cols[k] for k in ("ax", "ay", "az")]).clone())
            del dp
        assert torch.equal(outs[0], outs[1]), force
        a = outs[0]
        assert torch.isfinite(a).all() and (a != 0).any(), force
        net = (a * m).sum(dim=1).abs()
        handed = (a.abs() * m).sum(dim=1)
        assert (net <= 1e-10 * handed.max()).all(), (force, net.tolist(), handed.tolist())


def test_gr_full_sharded_driver_on_the_hip_engine(ctx):
    """The multi-GPU driver (reboundx_amd/dist.py) with no process group: all rows on this GPU,
    through the same engine calls a rank makes for its shard."""
    from reboundx_amd.device import DeviceParticles
    from reboundx_amd.dist import HipGrFullEngine, grfull_sharded
    n = 777
    cl = cluster_with_gravity(n)
    ref = oracle.Sim(cl)
    sweeps_ref = oracle.gr_full(ref, 1e2, 1)
    dp = DeviceParticles.from_host(cl)
    sweeps = grfull_sharded(HipGrFullEngine(ctx, dp, EXACT), 1e2, 1)
    assert sweeps == sweeps_ref
    assert ulp_diff(np.array(dp.acc_host()), np.array(ref.acc)).max() == 0


def test_multi_gpu_drivers_over_a_one_rank_rccl_group():
    """torch.distributed backend "nccl" (RCCL) on this GPU: the sharded gr_full driver and the
    sharded per-particle driver make the same collective calls as on several GPUs and must
    reproduce the runs made without a process group (child process: tests/rccl_one_rank.py)."""
    import os
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_one_rank.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-u", script], capture_output=True, text=True, timeout=110, env=env)
    assert r.returncode == 0 and "rccl one-rank ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_two_ranks_sharing_this_gpu_run_the_sharded_drivers(ctx, tmp_path):
    """Two processes (gloo; collectives staged through the host) run the row-sharded gr_full driver
    and the particle-sharded per-particle driver, including gr and Jacobi modify_orbits_forces with
    massive slices, against the one-process evaluation on the same GPU (child processes:
    tests/gloo_two_ranks.py)."""
    import os
    import subprocess
    import sys
    from reboundx_amd.device import DeviceParticles
    from reboundx_amd.dist import HipGrFullEngine, grfull_sharded
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gloo_two_ranks.py")
    port = 29700 + os.getpid() % 200
    outs = [str(tmp_path / ("rank%d.npz" % r)) for r in range(2)]
    procs = [subprocess.Popen([sys.executable, "-u", script, str(r), "2", str(port), outs[r]], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    logs = [p.communicate(timeout=140)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
    res = [np.load(o) for o in outs]
    # gr_full: every rank ends with all rows; against the unsharded run on this GPU
    cl = synthetic.cluster(1500)
    for k, s in (("ax", "x"), ("ay", "y"), ("az", "z")):
        cl[k] = -cl[s]
    dp = DeviceParticles.from_host(cl, n_head=1500, tail_massless=False)
    sweeps = grfull_sharded(HipGrFullEngine(ctx, dp, EXACT), 1e2, 0)
    one = np.array(dp.acc_host())
    base = np.stack([cl["ax"], cl["ay"], cl["az"]])
    for r in res:
        assert int(r["sweeps"]) == sweeps
        assert rel_err(r["acc"] - base, one - base).max() <= 1e-12
    assert np.array_equal(res[0]["acc"], res[1]["acc"])
    # per-particle stack: assemble the slices, compare with the unsharded device run
    n, h = 20001, 2
    cols, pp, _ = synthetic.disk(n, h - 1, dust_mass=1e-12)
    pp["primary"] = np.zeros(n, dtype=np.uint8); pp["primary"][0] = 1
    stack = [("tides_precession", {}), ("modify_orbits_forces", {}), ("gr_potential", {"c": C}), ("gr", {"c": C})]
    whole, _ = run_device(ctx, cols, pp, stack, 1.0, EXACT)
    got = np.zeros((3, n))
    for r in res:
        got[:, r["idx"][h:]] = r["slice_acc"][:, h:]
    assert np.array_equal(res[0]["slice_acc"][:, :h], res[1]["slice_acc"][:, :h])
    got[:, :h] = res[0]["slice_acc"][:, :h]
    big = np.abs(whole).max(axis=0)
    assert (np.abs(got - whole) / big[None, :]).max() <= 1e-12


@pytest.mark.parametrize("mode", [EXACT, FAST])
def test_particle_sharded_run_with_massive_slices_matches_unsharded_checker(ctx, mode):
    """Config C3 on several devices: every grain has mass, so each shard's pushes on the star
    have to be summed over shards.  The shards are run one after the other on this GPU; the
    all-reduce is replaced by a plain sum of the per-shard totals."""
    import torch
    from reboundx_amd.device import DeviceParticles
    from reboundx_amd.dist import particle_shard
    from reboundx_amd.program import build_ops
    n, h, w = 40001, 2, 3
    cols, pp, _ = synthetic.disk(n, h - 1, dust_mass=1e-12)
    pp["primary"] = np.zeros(n, dtype=np.uint8); pp["primary"][0] = 1
    cols = zero_acc(cols)
    ref = run_oracle(cols, pp, C3_STACK)
    ops = build_ops(C3_STACK, pp)
    shards, sums = [], []
    for r in range(w):
        lo, hi = particle_shard(n, h, w, r)
        idx = np.concatenate([np.arange(h), np.arange(lo, hi)])
        host = {k: v[idx] for k, v in cols.items()}
        for k in ("R_tides", "k1", "tau_a", "tau_e", "tau_inc"):
            host[k] = pp[k][idx]
        dp = DeviceParticles.from_host(host, n_head=h, tail_massless=False)
        ctx.apply(dp.head_view(), ops, mode)
        s = torch.zeros(3 * len(ops), dtype=torch.float64, device="cuda")
        ctx.apply_targets(dp.view(), ops, h, s, mode)
        shards.append((dp, idx)); sums.append(s)
    total = torch.stack(sums).sum(dim=0)                      # what the all-reduce delivers to every rank
    got = np.zeros((3, n))
    heads = []
    for dp, idx in shards:
        ctx.add_pushes(dp.view(), ops, total)
        a = np.array(dp.acc_host())
        got[:, idx[h:]] = a[:, h:]
        heads.append(a[:, :h])
    for hd in heads[1:]:
        assert np.array_equal(hd, heads[0])                   # replicated head identical on every rank
    got[:, :h] = heads[0]
    if mode == EXACT:
        assert ulp_diff(got[:, h:], ref[:, h:]).max() == 0
    assert rel_err(got[:, 1:], ref[:, 1:]).max() <= TOL
    scale = max(np.abs(ref[:, 1:]).max() * 1e-12 * n, np.abs(ref[:, 0]).max())
    assert np.abs(got[:, 0] - ref[:, 0]).max() <= 1e-12 * scale


@pytest.mark.parametrize("coordinates", [0, 1])
def test_sharded_jacobi_and_barycentric_drag_with_massive_slices(coordinates):
    """modify_orbits_forces in Jacobi / barycentric coordinates with every grain massive, sharded
    over three "devices" (run one after the other on this GPU, one context each; the two
    all-gathers are replaced by handing every rank the stacked vectors).  Compared with the
    unsharded checker under the prefix-sum bar (see scan_check)."""
    import torch
    from reboundx_amd.device import Context, DeviceParticles
    from reboundx_amd.dist import mof_com_sharded, particle_shard
    n, h, w = 30011, 3, 3
    cols, pp, _ = synthetic.disk(n, h - 1, dust_mass=1e-9)
    cols = zero_acc(cols)
    ref = run_oracle(cols, pp, [("modify_orbits_forces", {"coordinates": coordinates})])
    ctxs = [Context(0) for _ in range(w)]
    shards = []
    for r in range(w):
        lo, hi = particle_shard(n, h, w, r)
        idx = np.concatenate([np.arange(h), np.arange(lo, hi)])
        host = {k: v[idx] for k, v in cols.items()}
        for k in ("tau_a", "tau_e", "tau_inc"):
            host[k] = pp[k][idx]
        shards.append((DeviceParticles.from_host(host, n_head=h, tail_massless=False), idx))
    z = lambda k: torch.zeros(k, dtype=torch.float64, device="cuda")
    # what the all-gathers would deliver: every rank's slice sums, then every rank's slice pushes
    sums = []
    for c, (dp, _) in zip(ctxs, shards):
        s = z(7); c.shard_sums(dp.view(), h, dp.n, s); sums.append(s)
    sums = torch.stack(sums)
    head = z(7); ctxs[0].shard_sums(shards[0][0].view(), 0, h, head)
    pushes = []
    for r, (c, (dp, _)) in enumerate(zip(ctxs, shards)):
        before = head + (sums.sum(dim=0) if coordinates == 1 else sums[:r].sum(dim=0))
        p = z(3); c.shard_mof_forces(dp.view(), coordinates, h, dp.n, before, p); pushes.append(p)
    pushes = torch.stack(pushes)
    got = np.zeros((3, n))
    heads = []
    for r, (c, (dp, idx)) in enumerate(zip(ctxs, shards)):
        calls = iter([(sums, r), (pushes, r)])
        mof_com_sharded(c, dp, coordinates, exchange=lambda vec: next(calls))
        a = np.array(dp.acc_host())
        got[:, idx[h:]] = a[:, h:]
        heads.append(a[:, :h])
    for hd in heads[1:]:
        assert np.array_equal(hd, heads[0])                   # the replicated head agrees bit for bit on every rank
    got[:, :h] = heads[0]
    for c in ctxs:
        c.close()
    m = cols["m"]
    react = float((m[1:] / m.sum() * np.abs(ref[:, 1:]).max(axis=0)).sum())
    den = np.maximum(np.abs(ref).max(axis=0), react)[None, :]
    den = np.where(den == 0., 1., den)
    # against the same sums evaluated unsharded on one device: only the grouping of the partial sums differs
    c1 = Context(0)
    c1.sequential_exact(False)          # (the sharded entry points are prefix sums: compared with the one-device prefix sums)
    one, _ = run_device(c1, cols, pp, [("modify_orbits_forces", {"coordinates": coordinates})], 1.0, EXACT)
    c1.close()
    assert (np.abs(got - one) / den).max() <= 1e-13
    # against the sequential checker: 1e-12 of the component plus the checker's own measured distance from
    # the binary128 value of the same formulas (see scan_check)
    force = ("modify_orbits_forces", {"coordinates": coordinates})
    hi, lo, seq_err = exact_and_sequential(cols, pp, force)
    big = np.abs(ref).max(axis=0)
    assert (np.abs(got - ref) <= np.maximum(TOL * np.abs(ref), ROUNDING * big[None, :]) + seq_err).all()
    assert (err_vs_exact(got, hi, lo) <= ROUNDING * big[None, :]).all()


class ThreadGather:
    """all-gather among threads that each play one rank on this GPU: every thread deposits its
    vector, all wait, every thread gets the stacked vectors and its rank."""

    def __init__(self, world):
        import threading
        self.world, self.slots, self.barrier = world, [None] * world, threading.Barrier(world)

    def exchange_for(self, rank):
        import torch

        def exchange(vec):
            torch.cuda.synchronize()
            self.slots[rank] = vec.detach().clone()
            self.barrier.wait()
            out = torch.stack(self.slots)
            self.barrier.wait()
            return out, rank
        return exchange


@pytest.mark.parametrize("mode", [EXACT, FAST])
@pytest.mark.parametrize("layout", ["planets-in-slices", "every-body-massive"])
def test_sharded_gr_with_massive_bodies_in_the_slices(layout, mode):
    """gr with massive bodies outside the replicated head, sharded over three "devices" (three
    threads on this GPU, one context each, all-gathers emulated among the threads).  Compared with
    the unsharded device evaluation (only the grouping of partial sums differs) and with the
    sequential checker under the prefix-sum bar."""
    import threading
    from reboundx_amd.device import Context, DeviceParticles
    from reboundx_amd.dist import gr_sharded, particle_shard
    h, w = 2, 3
    if layout == "planets-in-slices":
        n = 20011
        cols, pp, _ = synthetic.disk(n, h - 1)
        rng = np.random.default_rng(3)
        for i in rng.choice(np.arange(h, n), size=9, replace=False):
            cols["m"][i] = 10.0 ** rng.uniform(-6, -3)
    else:
        n = 3001
        cols, pp, _ = synthetic.disk(n, h - 1, dust_mass=1e-9)
    synthetic.add_newtonian(cols, n_head=h)
    force = [("gr", {"c": C})]
    ref = run_oracle(cols, pp, force)
    c1 = Context(0)
    c1.sequential_exact(False)          # (the sharded entry points are prefix sums: compared with the one-device prefix sums)
    one, _ = run_device(c1, cols, pp, force, 1.0, mode)
    c1.close()
    tg = ThreadGather(w)
    got = np.zeros((3, n))
    heads, errors = [None] * w, []

    def rank_main(r):
        try:
            lo, hi = particle_shard(n, h, w, r)
            idx = np.concatenate([np.arange(h), np.arange(lo, hi)])
            dp = DeviceParticles.from_host({k: v[idx] for k, v in cols.items()}, n_head=h, tail_massless=False)
            c = Context(0)
            gr_sharded(c, dp, C, 10, mode, exchange=tg.exchange_for(r))
            a = np.array(dp.acc_host())
            got[:, idx[h:]] = a[:, h:]
            heads[r] = a[:, :h]
            c.close()
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            tg.barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(w)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(100)
    assert not errors, errors
    for hd in heads[1:]:
        assert np.array_equal(hd, heads[0])                   # the replicated head agrees bit for bit on every rank
    got[:, :h] = heads[0]
    den = np.abs(ref - cols_acc(cols)).max(axis=0)[None, :]
    den = np.maximum(den, np.abs(ref - cols_acc(cols)).max() * 1e-9)
    assert (np.abs(got - one) / den).max() <= 1e-12
    # against the sequential checker: 1e-12 of the component (floored at 1e-3 of the body's largest) plus the
    # checker's own measured distance from the binary128 value of the same formulas (see scan_check)
    hi, lo, seq_err = exact_and_sequential(cols, pp, force[0])
    upd = ref - cols_acc(cols)
    big = np.abs(upd).max(axis=0)
    # (+ one rounding of the sum with the incoming Newtonian acceleration, 1e8 times the size of the update)
    assert (np.abs(got - ref) <= np.maximum(TOL * np.abs(upd), ROUNDING * big[None, :]) + seq_err + np.spacing(np.abs(ref))).all()


@pytest.mark.parametrize("stack", [[("tides_precession", {}), ("modify_orbits_forces", {}), ("gr_potential", {"c": C})],
                                   [("gr", {"c": C}), ("tides_precession", {}), ("modify_orbits_forces", {"coordinates": 1})]],
                         ids=["jacobi-drag", "gr+barycentric-drag"])
def test_sharded_driver_routes_com_drag_through_the_prefix_sum_passes(ctx, stack):
    """apply_forces_sharded with a stack that mixes streaming forces with gr or Jacobi/barycentric
    modify_orbits_forces on one rank (no process group): same result as the unsharded evaluation."""
    from reboundx_amd.device import DeviceParticles
    from reboundx_amd.dist import apply_forces_sharded
    from reboundx_amd.program import build_ops
    n, h = 5000, 2
    cols, pp, _ = synthetic.disk(n, h - 1, dust_mass=1e-10)
    cols = zero_acc(cols)
    host = dict(cols)
    for k in ("R_tides", "k1", "tau_a", "tau_e", "tau_inc"):
        host[k] = pp[k]
    dp = DeviceParticles.from_host(host, n_head=h, tail_massless=False)
    apply_forces_sharded(ctx, dp, build_ops(stack, pp), EXACT)
    got = np.array(dp.acc_host())
    one, _ = run_device(ctx, cols, pp, stack, 1.0, EXACT)
    big = np.abs(one).max(axis=0)
    assert (np.abs(got - one) / big[None, :]).max() <= 1e-13
