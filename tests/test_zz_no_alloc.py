"""rebxamd_apply_static -- the entry point that allocates nothing (its scratch is a fixed array in
the library's code object) -- against rebxamd_apply, through the C ABI.

Both run the same kernels in the same fixed summation order, so the bar is bit equality (the parity
of rebxamd_apply itself against the CPU checker is covered by test_gpu_parity.py; here one case is
also checked against the checker directly).  Every case makes two calls in a row, the second on a
smaller input, so that whatever the first call left in the shared scratch array would show.
"""
import ctypes

import numpy as np
import pytest
import torch

from gpu_common import TOL, rel_err, run_oracle
from reboundx_amd import _lib, synthetic
from reboundx_amd._lib import Op, lib
from reboundx_amd.device import Context, DeviceParticles
from reboundx_amd.program import build_ops

pytestmark = pytest.mark.gpu

C = synthetic.C_DEFAULT
C2_STACK = [("radiation_forces", {"c": C}), ("gravitational_harmonics", {}), ("gr", {"c": C})]
C3_STACK = [("tides_precession", {}), ("modify_orbits_forces", {"coordinates": 2}), ("gr_potential", {"c": C})]
C5_STACK = [("radiation_forces", {"c": C}), ("gravitational_harmonics", {}), ("gr_potential", {"c": C}),
            ("tides_precession", {}), ("modify_orbits_forces", {"coordinates": 2})]


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


def particles(n, n_planets, dust_mass):
    cols, pp, _ = synthetic.disk(n, n_planets, dust_mass=dust_mass)
    pp["primary"] = np.zeros(n, dtype=np.uint8)
    pp["primary"][0] = 1
    synthetic.add_newtonian(cols, n_head=1 + n_planets)
    host = dict(cols)
    for k in ("beta", "R_tides", "k1", "tau_a", "tau_e", "tau_inc"):
        if pp.get(k) is not None:
            host[k] = pp[k]
    return cols, pp, host


def apply_static(view, ops, mode, warnings=None):
    arr = (Op * max(len(ops), 1))(*ops)
    err = ctypes.create_string_buffer(512)
    rc = lib.rebxamd_apply_static(ctypes.byref(view), arr, len(ops), mode,
                                  ctypes.c_void_p(warnings.data_ptr()) if warnings is not None else None,
                                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), err, len(err))
    return rc, err.value.decode()


def both_ways(ctx, host, pp, stack, mode):
    ops = build_ops(stack, pp, 0)
    a = DeviceParticles.from_host(host)
    b = DeviceParticles.from_host(host)
    ctx.apply(a.view(), ops, mode)
    rc, msg = apply_static(b.view(), ops, mode)
    assert rc == _lib.OK, msg
    torch.cuda.synchronize()
    return np.array(a.acc_host()), np.array(b.acc_host())


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("stack,n_planets,dust_mass", [(C2_STACK, 2, 0.0), (C5_STACK, 0, 0.0), (C3_STACK, 0, 1e-15), (C5_STACK, 3, 1e-12)],
                         ids=["c2-massless-tail", "c5-massless-tail", "c3-all-massive", "c5-all-massive"])
def test_static_scratch_entry_point_matches_rebxamd_apply_on_two_calls_in_a_row(ctx, stack, n_planets, dust_mass, mode):
    for n in (150001, 4099):                 # the second call is the smaller one
        cols, pp, host = particles(n, n_planets, dust_mass)
        want, got = both_ways(ctx, host, pp, stack, mode)
        assert np.isfinite(got).all()
        assert np.array_equal(want.view(np.int64), got.view(np.int64)), "n=%d: differs from rebxamd_apply in %d words" % (
            n, int((want.view(np.int64) != got.view(np.int64)).sum()))


def test_static_scratch_entry_point_takes_no_device_memory(ctx):
    """Free device memory as the driver reports it (hipMemGetInfo) is the same before and after calls on growing
    inputs, all-massive layout included (its partial sums live in the fixed array)."""
    sizes = (4099, 150001, 1_200_001)
    prepared = []
    for n in sizes:
        cols, pp, host = particles(n, 0, 1e-14)
        prepared.append((DeviceParticles.from_host(host), build_ops(C3_STACK, pp, 0)))
    rc, msg = apply_static(prepared[0][0].view(), prepared[0][1], 0)          # first call on the device: address look-up
    assert rc == _lib.OK, msg
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for dp, ops in prepared:
        for mode in (0, 1):
            rc, msg = apply_static(dp.view(), ops, mode)
            assert rc == _lib.OK, msg
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0


@pytest.mark.parametrize("stack,n_planets,dust_mass", [(C2_STACK, 2, 0.0), (C3_STACK, 0, 1e-15)], ids=["c2-massless-tail", "c3-all-massive"])
def test_static_scratch_entry_point_under_stream_capture(ctx, stack, n_planets, dust_mass):
    """The call is captured into a HIP graph (capture mode global: a hipMalloc, hipFree or a wait for the stream inside
    it would fail the capture) and the graph replayed twice; the accelerations equal those of rebxamd_apply called
    twice.  The look-up of the array's address happens on the first call on the device, made outside the capture."""
    n = 60001
    cols, pp, host = particles(n, n_planets, dust_mass)
    ops = build_ops(stack, pp, 0)
    warm = DeviceParticles.from_host(host)
    rc, msg = apply_static(warm.view(), ops, 0)
    assert rc == _lib.OK, msg
    want = DeviceParticles.from_host(host)
    ctx.apply(want.view(), ops, 0)
    ctx.apply(want.view(), ops, 0)
    dp = DeviceParticles.from_host(host)
    before = np.array(dp.acc_host())
    word = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=torch.cuda.Stream()):
        rc, msg = apply_static(dp.view(), ops, 0, warnings=word)
    assert rc == _lib.OK, msg
    torch.cuda.synchronize()
    assert np.array_equal(np.array(dp.acc_host()), before)            # captured, not run
    graph.replay()
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(np.array(dp.acc_host()), np.array(want.acc_host()))


def test_static_scratch_entry_point_serves_host_threads_in_turn(ctx):
    """Several host threads call at the same time, all on ONE stream (which orders their work on the GPU), each with its
    own particles and warning word.  The host side of a call is serialised inside the library, so the launches of one
    call stay together in the stream: every call's warning bits land in its own word and every result equals
    rebxamd_apply's.  (Were the launches of two calls to interleave, one call's head pass would overwrite the tables
    the other's streaming pass is about to read.)"""
    import threading
    n = 20011
    cols, pp, host = particles(n, 1, 0.0)
    loud, calm = build_ops([("gr", {"c": 3.0})], pp, 0), build_ops([("gr", {"c": C})], pp, 0)
    ref = {}
    for name, ops in (("loud", loud), ("calm", calm)):
        a = DeviceParticles.from_host(host)
        ctx.apply(a.view(), ops, 0)
        ref[name] = (np.array(a.acc_host()), ctx.fetch_warnings())
    assert ref["loud"][1] != 0 and ref["calm"][1] == 0
    shared = torch.cuda.Stream()
    errors, results = [], []

    def worker(name, ops):
        try:
            with torch.cuda.stream(shared):
                mine = []
                for _ in range(25):
                    dp = DeviceParticles.from_host(host)
                    word = torch.zeros(1, dtype=torch.int32, device="cuda:0")
                    rc, msg = apply_static(dp.view(), ops, 0, warnings=word)
                    assert rc == _lib.OK, msg
                    mine.append((dp, word))
                results.append((name, mine))
        except Exception as e:              # noqa: BLE001
            errors.append((name, repr(e)))

    threads = [threading.Thread(target=worker, args=(nm, ops)) for nm, ops in (("loud", loud), ("calm", calm), ("loud", loud), ("calm", calm))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    shared.synchronize()
    torch.cuda.synchronize()
    assert not errors, errors
    for name, mine in results:
        for dp, word in mine:
            assert int(word.item()) == ref[name][1], (name, int(word.item()))
            assert np.array_equal(np.array(dp.acc_host()), ref[name][0])


def test_static_scratch_entry_point_against_the_checker(ctx):
    n = 30011
    cols, pp, host = particles(n, 0, 1e-12)
    ref = run_oracle(cols, pp, C3_STACK)
    ops = build_ops(C3_STACK, pp, 0)
    for rep in range(2):                     # the same scratch array serves both calls
        dp = DeviceParticles.from_host(host)
        rc, msg = apply_static(dp.view(), ops, 0)
        assert rc == _lib.OK, msg
        got = np.array(dp.acc_host())
        assert rel_err(got[:, 1:], ref[:, 1:]).max() <= TOL       # <= 1e-12 per component
        scale = max(np.abs(ref[:, 1:]).max() * 1e-12 * n, np.abs(ref[:, 0]).max())
        assert np.abs(got[:, 0] - ref[:, 0]).max() <= 1e-12 * scale  # the row that collects the pushes: relative to the summed terms


def test_static_scratch_entry_point_raises_warning_bits_in_the_callers_word(ctx):
    """gr at a small speed of light (slow inner iteration): whatever bits rebxamd_fetch_warnings reports
    after rebxamd_apply land in the caller's device word here."""
    n = 2000
    cols, pp, host = particles(n, 1, 0.0)
    stack = [("gr", {"c": 3.0})]
    ops = build_ops(stack, pp, 0)
    a = DeviceParticles.from_host(host)
    ctx.apply(a.view(), ops, 0)
    want = ctx.fetch_warnings()
    word = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    b = DeviceParticles.from_host(host)
    rc, msg = apply_static(b.view(), ops, 0, warnings=word)
    assert rc == _lib.OK, msg
    torch.cuda.synchronize()
    assert int(word.item()) == want


def test_static_scratch_entry_point_refuses_what_does_not_fit(ctx):
    n = 3000
    cols, pp, host = particles(n, 0, 1e-12)
    dp = DeviceParticles.from_host(host)
    before = np.array(dp.acc_host())
    for stack in ([("gr_full", {"c": C})], [("gr", {"c": C})], [("modify_orbits_forces", {})]):
        rc, msg = apply_static(dp.view(), build_ops(stack, pp, 0), 0)
        assert rc == _lib.ERR_UNSUPPORTED and "rebxamd_apply" in msg, (rc, msg)
    # more partial sums than the array holds: a view that claims 2e9 all-massive particles is refused
    # before any kernel is launched (nothing is read through the pointers)
    v = dp.view()
    v.n = 2_000_000_000
    v.n_head = v.n
    rc, msg = apply_static(v, build_ops(C3_STACK, pp, 0), 0)
    assert rc == _lib.ERR_UNSUPPORTED and "fixed scratch" in msg and "limit" in msg, (rc, msg)
    torch.cuda.synchronize()
    assert np.array_equal(before, np.array(dp.acc_host()))
    rc, msg = apply_static(dp.view(), build_ops(C3_STACK, pp, 0), 7)
    assert rc == _lib.ERR_ARG
