"""The two configurations whose full size the CPU checker cannot run whole -- gr_full at 65 536
bodies (4.3e9 pairs per pass) and the stacked tides_precession + modify_orbits_forces +
gr_potential at 1e8 massive grains -- checked at that size through the C ABI:

  * on SAMPLED rows against the checker (a row of these computations depends on all bodies only
    through quantities the device has already produced and that are themselves checked: the
    potential sums, the previous iterate), to the bars of the small-size tests: 1e-10 for gr_full,
    1e-12 per component for the streaming forces;
  * through size-independent properties: the stopping rule's votes, momentum handed out = momentum
    taken back.
"""
import numpy as np
import pytest
import torch

import oracle
from gpu_common import TOL, rel_err, record, unfloored_rel_err
from reboundx_amd import synthetic
from reboundx_amd.device import Context, DeviceParticles
from reboundx_amd.program import build_ops

pytestmark = pytest.mark.gpu

C = synthetic.C_DEFAULT


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


def test_gr_full_at_65536_bodies_on_sampled_rows(ctx):
    """BASELINE configuration 4 on one GPU, default arithmetic.  The phases run on all rows on the device
    (the entry points a row-sharded driver uses); the checker evaluates the same phases on 64 sampled
    rows from the same inputs: its own potential sums of ALL bodies (the one all-pairs pass the CPU does
    whole, on threads), and for each sweep the device's previous iterate.  Then the whole evaluation
    (rebxamd_grfull, sweeps queued ahead of the host) must land on the same iterate with the same count."""
    n = 65536
    cl = synthetic.cluster(n)
    for k, s in (("ax", "x"), ("ay", "y"), ("az", "z")):      # the incoming field: smooth and of the size of the Newtonian one
        cl[k] = -cl[s]
    c = 1.0e4
    dp = DeviceParticles.from_host(cl, n_head=n, tail_massless=False)
    view = dp.view()
    z = lambda *shape: torch.zeros(*shape, dtype=torch.float64, device="cuda:0")
    an, a0, ac = [z(n) for _ in range(3)], [z(n) for _ in range(3)], [z(n) for _ in range(3)]
    pot = z(n)
    ctx.grfull_newton(view, 0, an, a0)
    ctx.grfull_pot(view, c, 0, n, pot)
    ctx.grfull_const(view, c, 0, n, pot, ac)

    sim = oracle.Sim(cl)
    s = sim.s
    import ctypes
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    ref_pot = np.zeros(n)
    oracle.lib.orc_gr_full_pot(ctypes.byref(s), ctypes.c_double(c), 0, n, P(ref_pot))
    e = np.abs(pot.cpu().numpy() - ref_pot) / ref_pot
    assert e.max() <= 1e-12, e.max()

    ranges = [(0, 16), (n // 3, n // 3 + 16), (2 * n // 3, 2 * n // 3 + 16), (n - 16, n)]
    rows = np.concatenate([np.arange(a, b) for a, b in ranges])
    ref_ac = [np.zeros(n) for _ in range(3)]
    for a, b in ranges:
        oracle.lib.orc_gr_full_const(ctypes.byref(s), ctypes.c_double(c), a, b, P(ref_pot), *[P(q) for q in ref_ac])
    got_ac = np.array([q.cpu().numpy() for q in ac])
    err = rel_err(got_ac[:, rows], np.array(ref_ac)[:, rows])
    record("fullsize:gr_full:const:n=65536", floored=err.max(), unfloored=unfloored_rel_err(got_ac[:, rows], np.array(ref_ac)[:, rows]).max())
    assert err.max() <= 1e-10

    an_h = [q.cpu().numpy() for q in an]
    old, new = a0, [z(n) for _ in range(3)]
    flag = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    votes = []
    for k in range(10):
        flag.zero_()
        ctx.grfull_sweep(view, c, 0, n, an, old, ac, new, flag)
        old_h = [np.ascontiguousarray(q.cpu().numpy()) for q in old]
        ref_new = [np.zeros(n) for _ in range(3)]
        ref_vote = 0
        for a, b in ranges:
            ref_vote |= oracle.lib.orc_gr_full_sweep(ctypes.byref(s), ctypes.c_double(c), a, b, *[P(q) for q in an_h], *[P(q) for q in old_h],
                                                     *[P(q) for q in ref_ac], *[P(q) for q in ref_new])
        got = np.array([q.cpu().numpy() for q in new])
        err = rel_err(got[:, rows], np.array(ref_new)[:, rows])
        assert err.max() <= 1e-10, (k, err.max())
        vote = int(flag.item())
        votes.append(vote)
        if k == 0:
            assert vote == 1 and ref_vote != 0       # the first sweep starts from zero: every row moves
        old, new = new, old
        if not vote:
            break
    sweeps_by_phases = len(votes)
    assert 2 <= sweeps_by_phases <= 10
    final = np.array([q.cpu().numpy() for q in old])

    dq = DeviceParticles.from_host(cl, n_head=n, tail_massless=False)
    sweeps = ctx.grfull(dq.view(), c, 0, 0)
    assert sweeps == sweeps_by_phases
    total = np.array(dq.acc_host())
    incoming = np.array([cl["ax"], cl["ay"], cl["az"]])
    assert (np.abs(total - (incoming + final)) <= np.spacing(np.abs(total))).all()      # the same iterate, added to the incoming field
    assert ctx.fetch_warnings() & 2 == 0


def stacked(coords):
    return [("tides_precession", {}), ("modify_orbits_forces", {"coordinates": coords}), ("gr_potential", {"c": C})]


def test_stacked_tides_drag_gr_potential_at_1e8_massive_grains(ctx):
    """BASELINE configuration 3 whole on one GPU (16 columns x 0.8 GB): star + 1e8 grains of 1e-15 solar
    masses, generated on the device.  PARTICLE coordinates: behind the star every grain's update depends on
    the star and on itself only, so the checker run on (star + sampled grains copied back from the device)
    must reproduce the sampled rows; the star's row is the sum of 1e8 pushes, checked through the momentum
    it must cancel.  Jacobi coordinates (prefix sums over all 1e8 bodies): momentum, and on the sampled rows
    agreement with the PARTICLE result to the size of the difference between the two frames (the centre of
    mass of the bodies before a grain is within 1e-7 of the star)."""
    n = 100_000_000
    cols = synthetic.disk_device(n, "cuda:0", dust_mass=1e-15)
    dp = DeviceParticles(cols, n_head=1, tail_massless=False)
    sp = synthetic.star_params()
    m = cols["m"]
    starts = np.linspace(1, n - 2001, 8).astype(np.int64)
    pick = np.concatenate([[0]] + [np.arange(a, a + 2000) for a in starts])
    pick_t = torch.from_numpy(pick).cuda()
    sub = {k: cols[k][pick_t].cpu().numpy() for k in ("x", "y", "z", "vx", "vy", "vz", "m", "ax", "ay", "az")}
    subpp = {k: cols[k][pick_t].cpu().numpy() for k in ("R_tides", "k1", "tau_a", "tau_e", "tau_inc")}
    subpp["primary"] = np.zeros(len(pick), dtype=np.uint8)
    subpp["primary"][0] = 1
    results = {}
    for label, coords in (("particle", 2), ("jacobi", 0)):
        for k in ("ax", "ay", "az"):
            cols[k].zero_()
        ctx.apply(dp.view(), build_ops(stacked(coords), sp), 0)
        a = torch.stack([cols[k] for k in ("ax", "ay", "az")])
        assert bool(torch.isfinite(a).all())
        net = (a * m).sum(dim=1).abs()
        handed = (a.abs() * m).sum(dim=1)
        assert bool((net <= 1e-10 * handed.max()).all()), (label, net.tolist(), handed.tolist())
        results[label] = a[:, pick_t].cpu().numpy()
        del a
    sim = oracle.Sim(sub)
    oracle.apply_forces(sim, stacked(2), subpp)
    ref = np.array(sim.acc)
    err = rel_err(results["particle"][:, 1:], ref[:, 1:])
    record("fullsize:c3:particle:n=1e8", floored=err.max(), unfloored=unfloored_rel_err(results["particle"][:, 1:], ref[:, 1:]).max())
    assert err.max() <= TOL
    big = np.abs(ref[:, 1:]).max(axis=0)
    assert (np.abs(results["jacobi"][:, 1:] - ref[:, 1:]) <= 1e-6 * big[None, :]).all()
