"""Experiment (development aid): does walking the bodies in chunks that fit in the 256 MiB last-level cache let the passes of
the Jacobi-coordinate modify_orbits_forces find each other's data there?  Compares the one-shot evaluation with the same
evaluation made chunk by chunk, last chunk first, through the particle-shard entry points (totals of every chunk first; then
per chunk: per-body pass, then the pass that applies it).
usage: python scripts/jacobi_chunked.py [n] [reps] [chunk sizes, comma separated; 0 = the one-shot evaluation only]
(under rocprofv3 --kernel-trace --stats with ONE chunk size, the totals of the scan_* kernels / (2 * reps + 4) are the GPU time of one evaluation)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from reboundx_amd import synthetic  # noqa: E402
from reboundx_amd.device import Context, DeviceParticles  # noqa: E402
from reboundx_amd.program import build_ops  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
chunks = [int(c) for c in sys.argv[3].split(",")] if len(sys.argv) > 3 else [262144, 524288, 1048576, 2097152, 4194304]
ctx = Context(0)
cols, pp, _ = synthetic.disk(n, 0, dust_mass=1e-15)
host = dict(cols)
for k in ("tau_a", "tau_e", "tau_inc"):
    host[k] = pp[k]
dp = DeviceParticles.from_host(host, n_head=1, tail_massless=False)
del cols, host
view = dp.view()
ops = build_ops([("modify_orbits_forces", {"coordinates": 0})], {k: v[:1] for k, v in pp.items()})
dev = dp.cols["x"].device


def timed(fn):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def reset():
    for k in ("ax", "ay", "az"):
        dp.cols[k].zero_()


if chunks == [0]:
    t_one = timed(lambda: ctx.apply(view, ops, 1))
    print("one shot (exact arithmetic): %.1f us" % (t_one * 1e6), flush=True)
    t_one = timed(lambda: ctx.apply(view, ops, 1))
    print("one shot (exact arithmetic): %.1f us" % (t_one * 1e6), flush=True)
    sys.exit(0)
if len(chunks) > 1:
    reset()
    ctx.apply(view, ops, 1)
    want = torch.stack([dp.cols[k].clone() for k in ("ax", "ay", "az")])
    t_one = timed(lambda: ctx.apply(view, ops, 1))
    print("one shot (exact arithmetic): %.1f us" % (t_one * 1e6), flush=True)
    t_fast = timed(lambda: ctx.apply(view, ops, 0))
    print("one shot (default arithmetic): %.1f us" % (t_fast * 1e6), flush=True)

for chunk in chunks:
    bounds = list(range(0, n, chunk)) + [n]
    nc = len(bounds) - 1
    sums = torch.zeros(nc, 7, dtype=torch.float64, device=dev)
    push = torch.zeros(nc + 1, 3, dtype=torch.float64, device=dev)
    before = torch.zeros(nc, 7, dtype=torch.float64, device=dev)

    def run(order_back=True, phases=None):
        t = [time.perf_counter()]
        for c in range(nc):
            ctx.shard_sums(view, bounds[c], bounds[c + 1], sums[c])
        before[1:] = torch.cumsum(sums, dim=0)[:-1]
        if phases is not None:
            torch.cuda.synchronize(); t.append(time.perf_counter())
        if order_back:
            after = torch.zeros(3, dtype=torch.float64, device=dev)
            for c in range(nc - 1, -1, -1):
                ctx.shard_mof_forces(view, 0, bounds[c], bounds[c + 1], before[c] if c > 0 else None, push[c])
                ctx.shard_mof_apply(view, 0, bounds[c], bounds[c + 1], after)
                after = after + push[c]
        else:
            for c in range(nc):
                ctx.shard_mof_forces(view, 0, bounds[c], bounds[c + 1], before[c] if c > 0 else None, push[c])
            suffix = torch.flip(torch.cumsum(torch.flip(push[:nc], [0]), dim=0), [0])
            for c in range(nc - 1, -1, -1):
                ctx.shard_mof_apply(view, 0, bounds[c], bounds[c + 1], suffix[c + 1] if c + 1 < nc else torch.zeros(3, dtype=torch.float64, device=dev))
        if phases is not None:
            torch.cuda.synchronize(); t.append(time.perf_counter())
            phases.append([b - a for a, b in zip(t[:-1], t[1:])])

    if len(chunks) == 1:
        t_back = timed(run)
        t_back = timed(run)
        print("chunks of %d: %.1f us wall" % (chunk, t_back * 1e6))
        continue
    reset()
    run()
    got = torch.stack([dp.cols[k] for k in ("ax", "ay", "az")])
    err = ((got - want).abs().max() / want.abs().max()).item()
    t_back = timed(run)
    t_fwd = timed(lambda: run(False))
    ph = []
    run(True, ph)
    print("chunks of %8d (%3d): adjacent passes %.1f us, all per-body passes then all applies %.1f us   (totals %.1f us, rest %.1f us; max diff %.1e)" % (
        chunk, nc, t_back * 1e6, t_fwd * 1e6, ph[0][0] * 1e6, ph[0][1] * 1e6, err), flush=True)
