#!/usr/bin/env python3
"""Benchmark of the additional-force hot path on MI355X.

Workload (the configuration the headline metric is quoted on): synthetic debris disk,
star + 10^7 massless grains per GPU, radiation_forces + gravitational_harmonics (J2, J4) +
gr evaluated as ONE fused pass over device-resident fp64 columns.  A "step" is one such
force evaluation over every particle of the job.  With N GPUs each rank owns its own 10^7
grains (plus a replica of the star): weak scaling, no data-path communication.

Prints one JSON line (rank 0).  `value` = particle-force evaluations per second of the whole
job, inputs already resident in HBM.  `roofline` prices the streaming kernel against HBM
(algorithmic 112 B per particle: pos 24 + vel 24 + m 8 + beta 8 + acc 24 read + 24 written),
its time measured with two HIP events that bracket the timed steps on the launch stream (per step: the kernel plus the
one-wavefront head pass in front of it).  `cpu_baseline` times the CPU checker
(a single-threaded port of the reference algorithm) on the same particles on this host.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from reboundx_amd import synthetic  # noqa: E402
from reboundx_amd.device import Context, DeviceParticles  # noqa: E402
from reboundx_amd.program import build_ops  # noqa: E402

N_PER_GPU = 10_000_000
BYTES_PER_EVAL = 112            # fused radiation + J2/J4 + gr
HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak
FP64_PEAK_TFLOPS = 78.6         # 256 CU x 64 FMA/clk x 2 flop x 2.4 GHz (vector fp64)
C = synthetic.C_DEFAULT
STACK = [("radiation_forces", {"c": C}), ("gravitational_harmonics", {}), ("gr", {"c": C})]


def make_shard(rank, world, n_per_gpu):
    """Global disk of world*n_per_gpu particles; rank r gets [r*n, (r+1)*n) plus the star."""
    n_total = world * n_per_gpu
    cols, pp, idx = synthetic.disk(n_total, 0, lo=rank * n_per_gpu, hi=(rank + 1) * n_per_gpu)
    synthetic.add_newtonian(cols, n_head=1)
    return cols, pp, idx


def cpu_baseline(cols, pp, reps, threads, stack=None):
    """The CPU checker (a port of the reference algorithm) on the same particles, on `threads`
    host threads: the particles behind the star are cut into `threads` slices, every thread
    evaluates the stack on [star | its slice] (the ctypes calls release the interpreter lock).
    Returns (evals/s on all threads, seconds, evals/s of one thread alone on its slice)."""
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    n = len(cols["x"])
    bounds = np.linspace(1, n, threads + 1).astype(np.int64)
    sims = []
    for k in range(threads):
        idx = np.concatenate([[0], np.arange(bounds[k], bounds[k + 1])])
        sims.append((oracle.Sim({c: v[idx] for c, v in cols.items()}), {c: v[idx] for c, v in pp.items()}))

    stack = STACK if stack is None else stack

    def work(item):
        sim, p = item
        for _ in range(reps):
            oracle.apply_forces(sim, stack, p)

    t0 = time.perf_counter()
    work(sims[0])
    single = (len(sims[0][0].cols["x"]) - 1) * reps / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(work, sims))
    dt = time.perf_counter() - t0
    return (n - 1) * reps / dt, dt, single


def grfull_cpu_baseline(n, sweeps, rows=8192):
    """The CPU checker's gr_full phases (potential sums, constant part, one sweep: oracle/rebx_oracle.c, the reference's
    loops restated in O(N^2)) on `rows` of the n rows of the same cluster, on all host threads the box grants (OpenMP inside
    the checker), extrapolated to a whole evaluation with `sweeps` sweeps: every row costs the same n - 1 pairs per phase."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes
    import oracle
    cl = synthetic.cluster(n)
    for k, src in (("ax", "x"), ("ay", "y"), ("az", "z")):
        cl[k] = -cl[src]
    sim = oracle.Sim(cl)
    rows = min(rows, n)
    z = lambda: np.zeros(n)
    pot, ac, an, ao, out = z(), [z(), z(), z()], [cl["ax"].copy(), cl["ay"].copy(), cl["az"].copy()], [z(), z(), z()], [z(), z(), z()]
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    ref, c = ctypes.byref(sim.s), ctypes.c_double(1.0e4)
    times = {}
    t0 = time.perf_counter()
    oracle.lib.orc_gr_full_pot(ref, c, 0, rows, P(pot))
    times["pot"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    oracle.lib.orc_gr_full_const(ref, c, 0, rows, P(pot), *[P(a) for a in ac])
    times["const"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    oracle.lib.orc_gr_full_sweep(ref, c, 0, rows, *[P(a) for a in an], *[P(a) for a in ao], *[P(a) for a in ac], *[P(a) for a in out])
    times["sweep"] = time.perf_counter() - t0
    per_eval = (times["pot"] + times["const"] + sweeps * times["sweep"]) * n / rows
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
    return {"value": n * (n - 1.0) * (2 + sweeps) / per_eval, "unit": "pair-interactions/s", "cores": threads, "kind": "port",
            "sample": "the checker's three gr_full phases on %d of the %d rows (%.2f + %.2f + %.2f s for potential sums, constant part, one sweep), "
                      "scaled by %d/%d rows and to %d sweeps: %.0f s per evaluation" % (rows, n, times["pot"], times["const"], times["sweep"], n, rows, sweeps, per_eval)}


def load_profile(name):
    """A committed profile summary under profiles/ (counter passes cannot run inside this script's timed run)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:  # noqa: BLE001
        return None


def grfull_section(ctx, n, mode=0, dist=None, device="cuda:0"):
    """Secondary metric: gr_full pair-interactions/s at n bodies; rows sharded over the ranks
    of the job (all-gather of the new accelerations per sweep over RCCL), or one GPU."""
    from reboundx_amd.dist import HipGrFullEngine, grfull_sharded
    world = dist.get_world_size() if dist is not None else 1
    eng = failure = None
    try:
        cl = synthetic.cluster(n)
        dp = DeviceParticles.from_host(cl, device=device, n_head=n, tail_massless=False)
        # incoming accelerations: Newtonian gravity is part of REBOUND, not of this path; any smooth
        # field of the right size exercises the same arithmetic
        for k, s in (("ax", "x"), ("ay", "y"), ("az", "z")):
            dp.cols[k].copy_(-dp.cols[s])
        eng = HipGrFullEngine(ctx, dp, mode)
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        failure = "%s: %s" % (type(e).__name__, e)
    if dist is not None:
        # the ranks agree that every one of them is set up before the first collective of the evaluation
        ok = torch.tensor([0.0 if failure else 1.0], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) == 0.0 and failure is None:
            failure = "another rank failed to set up"
    if failure:
        return {"error": failure}

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    grfull_sharded(eng, 1.0e4, 0)                  # warm-up (allocates workspaces, sets up RCCL rings)
    sync()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        sweeps = grfull_sharded(eng, 1.0e4, 0)
    sync()
    dt = (time.perf_counter() - t0) / reps
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    pairs = n * (n - 1.0) * (2 + sweeps)
    flops = n * (n - 1.0) * (70 + 36 * sweeps)
    peak = FP64_PEAK_TFLOPS * world
    out = {"n": n, "n_gpus": world, "sweeps": sweeps, "ms_per_eval": dt * 1e3, "pair_interactions_per_s": pairs / dt,
           "sharding": "rows i over ranks; all-gather of a_new per sweep" if world > 1 else "single GPU",
           "roofline": {"bound": "fp64-valu", "achieved": flops / dt / 1e12, "peak": peak, "unit": "TFLOP/s",
                        "frac": flops / dt / 1e12 / peak}}
    # the clock the chip holds under these kernels, from a committed counter pass (GRBM_GUI_ACTIVE / 8 XCDs / duration of the
    # millisecond-long dispatches): the vector peak actually on offer is 78.6 TF x clock / 2.4 GHz
    prof = load_profile(CLOCK_PROFILE)
    if prof is not None:
        clocks = [k["effective_clock_GHz"] for name, k in prof.get("kernels", {}).items()
                  if "grfull_fast" in name and k.get("effective_clock_GHz") and k.get("duration_us", 0) >= 1000.]
        if clocks:
            clock = sum(clocks) / len(clocks)
            out["effective_clock_GHz"] = clock
            out["frac_at_held_clock"] = out["roofline"]["frac"] * 2.4 / clock
            out["clock_source"] = "profiles/%s (committed counter pass, not measured in this run)" % CLOCK_PROFILE
    return out


TRAFFIC_SOURCE = ("profiles/traffic.json (committed: TCC counters of separate rocprofv3 --pmc passes over this command, "
                  "2 x FETCH_SIZE + WRITE_SIZE per launch; not measured in this run)")
CLOCK_PROFILE = "r03_clock_valu.json"
WH_PROFILE = "r03_wh_valu.json"     # VALU instructions per wavefront and step of the batched stepping passes (scripts/pmc_wh.sh)
WH_BATCH = 32                       # the library's default steps per pass of a queued Wisdom-Holman run
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 4     # wave64 vector instructions per second: 1024 SIMDs, one instruction per 4 cycles at 2.4 GHz


def valu_issue_roofline(key, n, seconds_per_step):
    """Vector-issue roofline of a stepping pass that keeps its bodies in registers over a batch of steps.  The instruction
    count per wavefront (128 grains) and step is not measured by this run: it comes from the committed counter pass
    profiles/<WH_PROFILE> (SQ_INSTS_VALU / SQ_WAVES of a 32-step pass), and the figure is labelled so."""
    prof = load_profile(WH_PROFILE)
    if prof is None or key not in prof:
        return None
    per_wave_step = prof[key]["valu_per_wave_and_step"]
    waves = (n + 127) // 128
    achieved = per_wave_step * waves / seconds_per_step
    return {"bound": "fp64-valu-issue", "achieved": achieved / 1e9, "peak": VALU_ISSUE_PEAK / 1e9, "unit": "G wave-instructions/s",
            "frac": achieved / VALU_ISSUE_PEAK, "instructions_per_wave_and_step": per_wave_step,
            "instructions_source": "profiles/%s (committed counter pass%s; not measured in this run)" % (
                WH_PROFILE, ", taken at commit " + prof["commit"] if prof.get("commit") else "")}


def sampled_deviation(before, after, cols, pp, sample, stack, dt, steps):
    """Correctness check of the queued device-resident run (not part of any timed figure): the CPU checker's Wisdom-Holman
    restatement takes the star and the sampled grains from their state before the run through the same `steps` steps.
    Massless grains feel the star and not one another, and nothing they do reaches the star, so the sample's trajectory
    is the same alone as among all grains."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    sub = {k: (before[k].copy() if k in before else np.ascontiguousarray(v[sample])) for k, v in cols.items()}
    ref = oracle.Sim(sub)
    t0 = time.perf_counter()
    oracle.whfast_steps(ref, dt, steps, stack, {k: np.ascontiguousarray(v[sample]) for k, v in pp.items()})
    want = np.array([ref.cols[k] for k in ("x", "y", "z", "vx", "vy", "vz")])
    r = np.sqrt((want[:3] ** 2).sum(axis=0))
    dev = float((np.abs(after - want).max(axis=0) / np.maximum(r, 1.)).max())
    return {"grains": int(len(sample) - 1), "steps": int(steps), "max_deviation": dev, "bound": 2e-10, "within_bound": bool(dev <= 2e-10),
            "measure": "max |device - checker| over positions and velocities of the sampled grains after the queued run, each grain's divided by max(1, its distance from the star)",
            "checker_seconds": time.perf_counter() - t0}


def c5_section(ctx, n, steps, mode, device, check_sample=True):
    """Secondary metric: end-to-end device-resident run -- star + n massless grains, five stacked
    effects in the kick of a Wisdom-Holman step, particle and parameter columns resident in HBM.
    Reports the per-step latency with the host waiting for every step (one stream
    synchronisation per step, as a host-driven integrator would) and with the steps queued."""
    stack = [("radiation_forces", {"c": C}), ("gravitational_harmonics", {}), ("gr_potential", {"c": C}),
             ("tides_precession", {}), ("modify_orbits_forces", {})]
    cols, pp, _ = synthetic.disk(n + 1, 0)
    host = dict(cols)
    for k in ("beta", "R_tides", "k1", "tau_a", "tau_e", "tau_inc"):
        host[k] = pp[k]
    dp = DeviceParticles.from_host(host, device=device, n_head=1, tail_massless=True)
    ops = build_ops(stack, {k: v[:1] for k, v in pp.items()})
    view = dp.view()
    dt = 2 * np.pi / 100.0
    ctx.wh_steps(view, ops, dt, 20, mode)
    torch.cuda.synchronize()
    # one call and one wait per step, as a host-driven integrator would make them (arguments built once, the C entry
    # point called directly: what a C caller pays, without this script's per-call argument marshalling)
    import ctypes
    from reboundx_amd._lib import Op, lib
    arr = (Op * len(ops))(*ops)
    vref, stream = ctypes.byref(view), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    def check(rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %d (%s)" % (what, rc, lib.rebxamd_last_error(ctx._h).decode(errors="replace")))

    t0 = time.perf_counter()
    for _ in range(steps):
        check(lib.rebxamd_wh_steps(ctx._h, vref, arr, len(ops), dt, 1, mode, stream), "rebxamd_wh_steps")
        torch.cuda.current_stream().synchronize()
    synced = (time.perf_counter() - t0) / max(steps, 1)
    # a sample of grains is kept aside before the queued run: the checker takes (star + sample) through the same steps afterwards
    pick = torch.from_numpy(np.sort(np.concatenate([[0], 1 + np.random.default_rng(5).choice(n, size=min(256, n), replace=False)]))).to(device)
    state_cols = ("x", "y", "z", "vx", "vy", "vz")
    before = {k: dp.cols[k][pick].cpu().numpy() for k in state_cols}
    t0 = time.perf_counter()
    ctx.wh_steps(view, ops, dt, steps, mode)
    torch.cuda.synchronize()
    queued = (time.perf_counter() - t0) / steps
    after = np.array([dp.cols[k][pick].cpu().numpy() for k in state_cols])
    sample_check = None
    if check_sample:
        sample_check = sampled_deviation(before, after, cols, pp, pick.cpu().numpy(), stack, dt, steps)
    # the same with the run left open between the host's waits (rebxamd_wh_run_*: the trailing half drift of a step is merged
    # with the next step's leading one, as REBOUND's WHFast does with safe_mode = 0): one launch and one wait per step
    check(lib.rebxamd_wh_run_begin(ctx._h, vref, arr, len(ops), dt, mode, stream), "rebxamd_wh_run_begin")
    t0 = time.perf_counter()
    for _ in range(steps):
        check(lib.rebxamd_wh_run_steps(ctx._h, 1, stream), "rebxamd_wh_run_steps")
        torch.cuda.current_stream().synchronize()
    synced_open = (time.perf_counter() - t0) / max(steps, 1)
    check(lib.rebxamd_wh_run_end(ctx._h, stream), "rebxamd_wh_run_end")
    torch.cuda.synchronize()
    x = dp.cols["x"]
    finite = bool(torch.isfinite(x).all().item())
    # Queued steps run in batches (rebxamd_wh_batch, 32 by default): one pass takes every grain through the steps of a batch in
    # registers -- positions, velocities, beta and the three timescales read (80 B) and positions and velocities written
    # (48 B) once per BATCH, i.e. 4 B per grain and step -- so the pass is bound by fp64 vector issue, not by memory.
    # Instructions per wavefront (128 grains) and step: SQ_INSTS_VALU / SQ_WAVES of a 32-step pass (scripts/pmc_wh.sh 1000000 128),
    # read from the committed summary of that counter pass.
    # A wave64 vector instruction occupies its SIMD for 4 cycles (more for the quarter-rate reciprocals and square roots).
    out = {"n": n, "steps": steps, "effects": 5, "steps_per_pass": WH_BATCH, "us_per_step_host_synced": synced * 1e6,
           "us_per_step_host_synced_open_run": synced_open * 1e6, "us_per_step_queued": queued * 1e6,
           "particle_steps_per_s": n / queued, "state_finite": finite, "sampled_grains_vs_checker": sample_check,
           "hbm_bytes_per_particle_step": 128. / WH_BATCH,
           "roofline": valu_issue_roofline("five_effects", n, queued)}
    # the shape of BASELINE configuration 1 (star + planet + grains under radiation forces) at this size: the massive head steps in
    # Jacobi coordinates in one wavefront (three small launches per step), the grains in the same kind of pass, which also adds the
    # pull of both head bodies; positions, velocities and beta read (56 B), positions and velocities written (48 B)
    del dp
    steps1 = max(1, steps // 4)
    cols, pp, _ = synthetic.disk(n + 2, 1)
    host = dict(cols)
    host["beta"] = pp["beta"]
    dq = DeviceParticles.from_host(host, device=device, n_head=2, tail_massless=True)
    ops1 = build_ops([("radiation_forces", {"c": C})], {k: v[:2] for k, v in pp.items()})
    v1 = dq.view()
    ctx.wh_steps(v1, ops1, dt, 20, mode)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.wh_steps(v1, ops1, dt, steps1, mode)
    torch.cuda.synchronize()
    q1 = (time.perf_counter() - t0) / steps1
    # one call and one wait per step behind the planet as well: closed steps, then a run left open between the waits
    arr1 = (Op * len(ops1))(*ops1)
    v1ref = ctypes.byref(v1)
    steps2 = max(1, steps // 10)
    t0 = time.perf_counter()
    for _ in range(steps2):
        check(lib.rebxamd_wh_steps(ctx._h, v1ref, arr1, len(ops1), dt, 1, mode, stream), "rebxamd_wh_steps behind star + planet")
        torch.cuda.current_stream().synchronize()
    s1 = (time.perf_counter() - t0) / steps2
    check(lib.rebxamd_wh_run_begin(ctx._h, v1ref, arr1, len(ops1), dt, mode, stream), "rebxamd_wh_run_begin behind star + planet")
    t0 = time.perf_counter()
    for _ in range(steps2):
        check(lib.rebxamd_wh_run_steps(ctx._h, 1, stream), "rebxamd_wh_run_steps behind star + planet")
        torch.cuda.current_stream().synchronize()
    so1 = (time.perf_counter() - t0) / steps2
    check(lib.rebxamd_wh_run_end(ctx._h, stream), "rebxamd_wh_run_end behind star + planet")
    torch.cuda.synchronize()
    out["star_planet_grains"] = {"n": n, "steps": steps1, "effects": 1, "us_per_step_queued": q1 * 1e6, "particle_steps_per_s": n / q1,
                                 "us_per_step_host_synced": s1 * 1e6, "us_per_step_host_synced_open_run": so1 * 1e6,
                                 "state_finite": bool(torch.isfinite(dq.cols["x"]).all().item()),
                                 "steps_per_pass": WH_BATCH, "hbm_bytes_per_particle_step": 104. / WH_BATCH,
                                 # (radiation + the pull of both head bodies + the Jacobi term + the drift)
                                 "roofline": valu_issue_roofline("star_planet_radiation", n, q1)}
    return out


def c3_section(ctx, n, mode, device):
    """Secondary metric (BASELINE config 3 on this GPU's share): tides_precession + modify_orbits_forces +
    gr_potential with every grain massive, so the pushes on the star are parallel reductions; PARTICLE
    coordinates (one streaming pass) and the reference's default Jacobi coordinates (prefix-sum passes)."""
    cols, pp, _ = synthetic.disk(n, 0, dust_mass=1e-15)
    pp["primary"] = np.zeros(n, dtype=np.uint8)
    pp["primary"][0] = 1
    host = dict(cols)
    for k in ("R_tides", "k1", "tau_a", "tau_e", "tau_inc"):
        host[k] = pp[k]
    dp = DeviceParticles.from_host(host, device=device, n_head=1, tail_massless=False)
    view = dp.view()
    out = {"n": n, "bytes_per_eval": 144}
    for label, coords in (("particle", 2), ("jacobi", 0)):
        stack = [("tides_precession", {}), ("modify_orbits_forces", {"coordinates": coords}), ("gr_potential", {"c": C})]
        ops = build_ops(stack, {k: v[:1] for k, v in pp.items()})
        for _ in range(3):
            ctx.apply(view, ops, mode)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            ctx.apply(view, ops, mode)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        out[label] = {"us_per_eval": dt * 1e6, "evals_per_s": n / dt,
                      "roofline": {"bound": "hbm", "achieved": 144 * n / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 144 * n / dt / 1e9 / HBM_PEAK_GBS}}
        if label == "jacobi":
            # The roofline above prices the stack's 144 algorithmic bytes.  Jacobi coordinates tie every body to the sums over the
            # bodies before it and to the reactions of the bodies behind it: a forward and a backward sweep over data that does
            # not fit on the chip cannot move less than 240 B per body, and the three sweeps of this implementation move 280 B.
            out[label]["bytes_moved_per_body"] = 280
            out[label]["two_sweep_floor_bytes_per_body"] = 240
            out[label]["frac_of_peak_on_bytes_moved"] = 280 * n / dt / 1e9 / HBM_PEAK_GBS
    return out


def massive_stack_cpu_baseline(n, threads, reps=200):
    """The CPU checker on the all-massive stack in PARTICLE coordinates (tides_precession + modify_orbits_forces +
    gr_potential; the star collects the pushes), n grains of the same synthetic disk sliced over `threads` host threads."""
    cols, pp, _ = synthetic.disk(n, 0, dust_mass=1e-15)
    pp["primary"] = np.zeros(n, dtype=np.uint8)
    pp["primary"][0] = 1
    stack = [("tides_precession", {}), ("modify_orbits_forces", {"coordinates": 2}), ("gr_potential", {"c": C})]
    v, secs, single = cpu_baseline(cols, pp, reps, threads, stack)
    return {"value": v, "unit": "evals/s", "cores": threads, "kind": "port",
            "sample": "%d evaluations of a %d-grain slice of the same disk (PARTICLE coordinates), grains sliced over %d threads (%.1f s); "
                      "one thread alone: %.3g evals/s" % (reps, n, threads, secs, single)}


def timed_evals(ctx, view, ops, mode, reps, warm=2):
    """Seconds per rebxamd_apply (wall clock around `reps` queued evaluations) and the HIP-event time of its
    streaming passes per evaluation."""
    for _ in range(warm):
        ctx.apply(view, ops, mode)
    torch.cuda.synchronize()
    ctx.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.apply(view, ops, mode)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    kernel_ms, _ = ctx.timing_read()
    ctx.timing_enable(False)
    return dt, kernel_ms / 1e3 / reps


def ooc_section(ctx, n, mode, device):
    """The headline stack on star + n massless grains with n large enough that nothing survives in the 256 MiB
    last-level cache from one evaluation to the next (4e7 grains: 3.5 GB of columns): the same kernel as the
    headline line, priced the same way (112 B per grain against 8 TB/s, HIP events on the launch stream)."""
    cols = synthetic.disk_device(n + 1, device, dust_mass=0.0, newtonian=True)
    keep = ("x", "y", "z", "vx", "vy", "vz", "m", "ax", "ay", "az", "beta")
    dp = DeviceParticles({k: cols[k] for k in keep}, n_head=1, tail_massless=True)
    del cols
    ops = build_ops(STACK, synthetic.star_params())
    dt, kernel_s = timed_evals(ctx, dp.view(), ops, mode, 10)
    achieved = BYTES_PER_EVAL * n / kernel_s / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("out_of_cache", {}).get("hbm_bytes_per_launch")
        except Exception:  # noqa: BLE001
            traffic = None
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": TRAFFIC_SOURCE if traffic is not None else None, "particles": n, "kernel": "stream_kernel (tail pass)", "kernel_us": kernel_s * 1e6, "us_per_eval_wall": dt * 1e6,
            "algorithmic_bytes_per_launch": BYTES_PER_EVAL * n,
            "note": "same kernel and stack as the headline line; at 1e7 grains the three acceleration columns (240 MB) stay in the last-level cache between evaluations, here nothing does"}


def c3_big_setup(n_total, device, rank, world):
    """BASELINE configuration 3 at its full size: tides_precession + modify_orbits_forces + gr_potential on n_total
    grains of 1e-15 solar masses (every body massive), generated on the device.  One GPU: the whole disk through
    rebxamd_apply.  Several: the grains sharded by index over the ranks (every rank holds [star | its slice]); the
    pushes on the star are summed with one all-reduce of 3 doubles per force, and in Jacobi coordinates the prefix
    and suffix sums cross ranks through two all-gathers of 7 and 3 doubles (reboundx_amd.dist).  Strong scaling: the
    total size is fixed; the time is the slowest rank's, collectives included."""
    from reboundx_amd.dist import particle_shard
    lo, hi = particle_shard(n_total, 1, world, rank)
    cols = synthetic.disk_device(1 + (hi - lo), device, dust_mass=1e-15, seed=3 + rank)
    keep = ("x", "y", "z", "vx", "vy", "vz", "m", "ax", "ay", "az", "R_tides", "k1", "tau_a", "tau_e", "tau_inc")
    dp = DeviceParticles({k: cols[k] for k in keep}, n_head=1, tail_massless=False)
    del cols
    sp = synthetic.star_params()
    return dp, sp


def c3_big_section(ctx, n_total, mode, dist, device, rank, world, red_dev):
    # Every rank sets up its slice first; the ranks then agree (one all-reduce of a flag) whether all of them succeeded,
    # and only then enter the timed collectives: a rank that failed alone (out of memory at 1e8 bodies, say) must not
    # leave the others waiting at a barrier.
    dp = sp = None
    failure = None
    try:
        dp, sp = c3_big_setup(n_total, device, rank, world)
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        failure = "%s: %s" % (type(e).__name__, e)
    if dist is not None:
        ok = torch.tensor([0.0 if failure else 1.0], dtype=torch.float64, device=red_dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) == 0.0 and failure is None:
            failure = "another rank failed to set up its slice"
    if failure:
        return {"error": failure}
    from reboundx_amd.dist import apply_forces_sharded
    out = {"n_total": n_total, "n_gpus": world, "bytes_per_eval": 144, "scaling": "strong",
           "sharding": "one GPU, rebxamd_apply" if world == 1 else "grains by index over ranks; all-reduce of the pushes on the star, all-gathers of the prefix/suffix sums"}
    reps = 5
    for label, coords in (("particle", 2), ("jacobi", 0)):
        stack = [("tides_precession", {}), ("modify_orbits_forces", {"coordinates": coords}), ("gr_potential", {"c": C})]
        ops = build_ops(stack, sp)

        def run():
            if world == 1:
                ctx.apply(dp.view(), ops, mode)
            else:
                apply_forces_sharded(ctx, dp, ops, mode)
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = (time.perf_counter() - t0) / reps
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        peak = HBM_PEAK_GBS * world
        out[label] = {"ms_per_eval": dt * 1e3, "evals_per_s": n_total / dt,
                      "roofline": {"bound": "hbm", "achieved": 144 * n_total / dt / 1e9, "peak": peak, "unit": "GB/s", "frac": 144 * n_total / dt / 1e9 / peak}}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n-per-gpu", type=int, default=N_PER_GPU)
    ap.add_argument("--mode", choices=["fast", "exact"], default="fast")
    ap.add_argument("--cpu-reps", type=int, default=12)
    ap.add_argument("--cpu-threads", type=int, default=16, help="host threads of the CPU baseline (the GPU box grants 16)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--grfull-n", type=int, default=65536, help="0 disables the gr_full section")
    ap.add_argument("--c5-n", type=int, default=1_000_000, help="0 disables the device-resident stepping section")
    ap.add_argument("--c5-steps", type=int, default=10_000)
    ap.add_argument("--c3-n", type=int, default=10_000_000, help="0 disables the all-massive stack section")
    ap.add_argument("--c3-big-n", type=int, default=100_000_000, help="total grains of the full-size all-massive stack, sharded over the ranks; 0 disables")
    ap.add_argument("--ooc-n", type=int, default=40_000_000, help="grains of the out-of-cache run of the headline stack; 0 disables")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus must equal WORLD_SIZE")
    dist = None
    # REBXAMD_DIST_BACKEND=gloo rehearses a multi-rank launch on a box with fewer GPUs than ranks (ranks share
    # devices; RCCL refuses that).  The driver's runs use the default: nccl = RCCL, one rank per GPU.
    backend = os.environ.get("REBXAMD_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    red_dev = "cuda" if backend == "nccl" else "cpu"       # where the two scalar reductions of the timing live

    def barrier():
        if dist is not None:
            dist.barrier()

    mode = 0 if args.mode == "fast" else 1
    cols, pp, idx = make_shard(rank, world, args.n_per_gpu)
    host = dict(cols)
    host["beta"] = pp["beta"]
    dp = DeviceParticles.from_host(host, device="cuda:%d" % local_rank, n_head=1, tail_massless=True)
    ops = build_ops(STACK, {k: v[:1] for k, v in pp.items()})      # every source is the star
    ctx = Context(local_rank)
    view = dp.view()
    n_local = dp.n

    for _ in range(args.warmup):
        ctx.apply(view, ops, mode)
    torch.cuda.synchronize()
    barrier()
    # Two HIP events bracket the K timed steps on the launch stream (torch's current stream is the one the library launches
    # on): device time per step = the streaming kernel + the one-wavefront head pass in front of it, an upper bound of the
    # kernel's own duration.  (Events around every launch inside the loop cost 3-6 us per step, on the wall clock and in
    # the durations they report alike: `scripts/bench_stream.py` and the rocprofv3 trace under profiles/ give the kernel alone.)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        ctx.apply(view, ops, mode)
    ev1.record()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms, launches = ev0.elapsed_time(ev1), args.steps

    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        cnt = torch.tensor([n_local], dtype=torch.float64, device=red_dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        n_job = float(cnt.item())
    else:
        n_job = float(n_local)

    if rank == 0:
        kernel_s = kernel_ms / 1e3 / max(launches, 1)
        tail = n_local - 1
        achieved = BYTES_PER_EVAL * tail / kernel_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:  # noqa: BLE001
                traffic = None
        out = {
            "metric": "particle-force evals/s (fp64) at 10^7 particles",
            "value": n_job * args.steps / dt,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "synthetic debris disk: star + 1e7 massless grains per GPU; radiation_forces + "
                                   "gravitational_harmonics(J2,J4) + gr fused, device-resident fp64 columns",
                       "particles_per_gpu": args.n_per_gpu, "arithmetic": args.mode, "sharding": "particle index, no communication"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": TRAFFIC_SOURCE if traffic is not None else None, "kernel": "stream_kernel (tail pass; the HIP-event time per step also holds the one-wavefront head pass launched in front of it)",
                         "kernel_us": kernel_s * 1e6,
                         "algorithmic_bytes_per_launch": BYTES_PER_EVAL * tail},
        }
        if not args.no_cpu and world == 1:
            threads = max(1, min(args.cpu_threads, os.cpu_count() or 1))
            v, secs, single = cpu_baseline(cols, pp, args.cpu_reps, threads)
            out["cpu_baseline"] = {"value": v, "unit": "evals/s", "cores": threads, "kind": "port",
                                   "sample": "%d evaluations of the same %d-particle disk, particles sliced over %d threads (%.1f s); "
                                             "one thread alone: %.3g evals/s" % (args.cpu_reps, n_local, threads, secs, single)}
    gr = None
    if args.grfull_n > 0:
        try:
            gr = grfull_section(ctx, args.grfull_n, mode, dist, "cuda:%d" % local_rank)     # collective: every rank takes part
        except Exception as e:  # noqa: BLE001 -- a failure of this secondary section must not lose the headline line
            gr = {"error": "%s: %s" % (type(e).__name__, e)}
    big = None
    if args.c3_big_n > 0:
        try:
            big = c3_big_section(ctx, args.c3_big_n, mode, dist, "cuda:%d" % local_rank, rank, world, red_dev)    # collective
        except Exception as e:  # noqa: BLE001
            big = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0:
        if big is not None:
            out["massive_stack_full_size"] = big
        if args.ooc_n > 0:
            try:
                out["roofline_out_of_cache"] = ooc_section(ctx, args.ooc_n, mode, "cuda:%d" % local_rank)
            except Exception as e:  # noqa: BLE001
                out["roofline_out_of_cache"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if args.c5_n > 0:
            try:
                out["resident_run"] = c5_section(ctx, args.c5_n, args.c5_steps, mode, "cuda:%d" % local_rank, check_sample=not args.no_cpu)
            except Exception as e:  # noqa: BLE001
                out["resident_run"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if args.c3_n > 0:
            try:
                out["massive_stack"] = c3_section(ctx, args.c3_n, mode, "cuda:%d" % local_rank)
            except Exception as e:  # noqa: BLE001
                out["massive_stack"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if gr is not None:
            out["gr_full"] = gr
        if not args.no_cpu and world == 1:
            threads = max(1, min(args.cpu_threads, os.cpu_count() or 1))
            if gr is not None and "error" not in gr:
                try:
                    gr["cpu_baseline"] = grfull_cpu_baseline(args.grfull_n, gr["sweeps"])
                except Exception as e:  # noqa: BLE001
                    gr["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
            ms = out.get("massive_stack")
            if ms is not None and "error" not in ms:
                try:
                    ms["cpu_baseline"] = massive_stack_cpu_baseline(min(args.c3_n, 1_000_000), threads)
                except Exception as e:  # noqa: BLE001
                    ms["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
        print(json.dumps(out), flush=True)
    barrier()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
