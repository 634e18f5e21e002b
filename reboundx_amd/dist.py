"""Multi-GPU drivers: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI
on the GPUs; "gloo" on CPU for tests).

* Per-particle forces shard by particle index with no communication: every rank holds the
  head (the massive bodies, replicated) followed by its own slice of the massless particles
  (:func:`particle_shard`).
* gr_full shards the rows i of the all-pairs sums.  Every rank holds all positions,
  velocities and masses (80 B per body; 5 MB at N = 65 536); per phase it computes its rows
  and the rows are all-gathered: once for the potential sums, once per substitution sweep for
  the new accelerations, together with the one-word "not yet converged" vote.

The driver is written against a small engine interface so that the same code runs on the HIP
library (:class:`HipGrFullEngine`) and, in tests, on the CPU checker.
"""
import torch


def _all_gather_into(out, inp, group=None):
    """dist.all_gather_into_tensor; device tensors under the gloo backend (rehearsals with several
    ranks on one GPU, where RCCL refuses to run) are staged through the host."""
    import torch.distributed as dist
    if out.is_cuda and dist.get_backend(group) == "gloo":
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, inp.contiguous().cpu(), group=group)
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, inp.contiguous(), group=group)


def row_shard(n, world, rank):
    """Contiguous, equal-size (padded) row ranges.  Returns (row0, row1, rows_per_rank)."""
    per = (n + world - 1) // world
    r0 = min(rank * per, n)
    return r0, min(r0 + per, n), per


def particle_shard(n_total, n_head, world, rank):
    """Global index range [lo, hi) of the tail particles owned by `rank` (head excluded)."""
    tail = n_total - n_head
    per = (tail + world - 1) // world
    lo = n_head + min(rank * per, tail)
    return lo, min(lo + per, n_total)


def apply_forces_sharded(ctx, particles, ops, mode=0, group=None):
    """One evaluation of per-particle forces when `particles` holds [replicated head | this
    rank's slice] and the slice may be massive: the head is updated redundantly on every rank,
    the slice locally, and each op's push on its source particle is summed over ranks with one
    small all-reduce (3 doubles per op)."""
    import torch.distributed as dist
    from ._lib import OP_GR, OP_MOF
    scans = [o for o in ops if (o.kind == OP_MOF and o.iarg in (0, 1)) or o.kind == OP_GR]
    if scans:
        # gr and Jacobi/barycentric modify_orbits_forces are prefix/suffix sums over the global order: they
        # run as passes of their own (no op reads accelerations, so only the order of additions changes)
        rest = [o for o in ops if o not in scans]
        sums = apply_forces_sharded(ctx, particles, rest, mode, group) if rest else None
        for o in scans:
            if o.kind == OP_GR:
                gr_sharded(ctx, particles, o.a, o.iarg, mode, group)
            else:
                mof_com_sharded(ctx, particles, o.iarg, group)
        return sums
    view = particles.view()
    ctx.apply(particles.head_view(), ops, mode)                   # head on head, with its own back-reactions
    sums = torch.zeros(3 * len(ops), dtype=torch.float64, device=particles.cols["x"].device)
    ctx.apply_targets(view, ops, particles.n_head, sums, mode)    # forces on the slice; pushes -> sums
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if sums.is_cuda and dist.get_backend(group) == "gloo":      # rehearsal on a shared GPU: through the host
            host = sums.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            sums.copy_(host)
        else:
            dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    ctx.add_pushes(view, ops, sums)
    return sums


def mof_com_sharded(ctx, particles, coordinates, group=None, exchange=None):
    """modify_orbits_forces in Jacobi (0) or barycentric (1) coordinates on [replicated head |
    this rank's slice] with massive slices: two tiny all-gathers (7 and 3 doubles per rank) carry
    the prefix sums and the back-reactions across ranks.  `exchange(vec)` -> (world, len) tensor
    and rank, overriding torch.distributed (used by tests that run the ranks one after the other)."""
    import torch.distributed as dist
    view = particles.view()
    K, n = particles.n_head, particles.n
    dev = particles.cols["x"].device

    def gather(vec):
        if exchange is not None:
            return exchange(vec)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            world, rank = dist.get_world_size(group), dist.get_rank(group)
            out = torch.empty(world * vec.numel(), dtype=vec.dtype, device=vec.device)
            _all_gather_into(out, vec, group)
            return out.view(world, -1), rank
        return vec.view(1, -1), 0

    z = lambda k: torch.zeros(k, dtype=torch.float64, device=dev)
    head_sums, slice_sums = z(7), z(7)
    ctx.shard_sums(view, 0, K, head_sums)
    ctx.shard_sums(view, K, n, slice_sums)
    all_sums, rank = gather(slice_sums)
    bary = coordinates == 1
    if bary:
        before_head = before_slice = head_sums + all_sums.sum(dim=0)
    else:
        before_head, before_slice = None, head_sums + all_sums[:rank].sum(dim=0)
    head_push, slice_push = z(3), z(3)
    ctx.shard_mof_forces(view, coordinates, 0, K, before_head, head_push)
    ctx.shard_mof_forces(view, coordinates, K, n, before_slice, slice_push)
    all_push, rank = gather(slice_push)
    if bary:
        after_head = after_slice = head_push + all_push.sum(dim=0)
    else:
        after_head, after_slice = all_push.sum(dim=0), all_push[rank + 1:].sum(dim=0)
    ctx.shard_mof_apply(view, coordinates, 0, K, after_head)
    ctx.shard_mof_apply(view, coordinates, K, n, after_slice)


def _gatherer(group, exchange):
    """all-gather of equal-length device vectors -> ((world, len) tensor, rank).  `exchange`
    overrides torch.distributed (tests that run the ranks one after the other)."""
    import torch.distributed as dist

    def gather(vec):
        if exchange is not None:
            return exchange(vec)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            world, rank = dist.get_world_size(group), dist.get_rank(group)
            out = torch.empty(world * vec.numel(), dtype=vec.dtype, device=vec.device)
            _all_gather_into(out, vec, group)
            return out.view(world, -1), rank
        return vec.reshape(1, -1), 0
    return gather


def gr_sharded(ctx, particles, c, max_iterations=10, mode=0, group=None, exchange=None, slice_offsets=None):
    """gr (reference src/gr.c:73-171) on [replicated head | this rank's slice] with massive bodies in
    the slices.  Four small all-gathers: the number of massive bodies per slice, their records
    (padded to the largest count), the 10 prefix sums and the 3 back-reaction sums per slice.
    `slice_offsets[r]`: number of slice bodies on the ranks before r (default: from the gathered
    slice lengths)."""
    view = particles.view()
    K, n = particles.n_head, particles.n
    dev = particles.cols["x"].device
    gather = _gatherer(group, exchange)
    z = lambda k: torch.zeros(k, dtype=torch.float64, device=dev)

    # 1. massive bodies: head (replicated, same on every rank) and slices
    lens, rank = gather(torch.tensor([float(n - K)], dtype=torch.float64, device=dev))
    lens = [int(v) for v in lens[:, 0].tolist()]
    if slice_offsets is None:
        slice_offsets = [sum(lens[:r]) for r in range(len(lens))]
    id_head, id_slice = 0, K + slice_offsets[rank]
    head_rec, head_cnt = z(5 * max(K, 1)), z(1)
    ctx.shard_massive(view, 0, K, id_head, head_rec, head_cnt)
    slice_rec, slice_cnt = z(5 * max(n - K, 1)), z(1)
    ctx.shard_massive(view, K, n, id_slice, slice_rec, slice_cnt)
    counts, _ = gather(slice_cnt)
    counts = [int(v) for v in counts[:, 0].tolist()]
    cap = max(max(counts), 1)
    padded = z(5 * cap)
    padded[:5 * counts[rank]].copy_(slice_rec[:5 * counts[rank]])
    all_rec, _ = gather(padded)
    nh = int(head_cnt.item())
    records = torch.cat([head_rec[:5 * nh]] + [all_rec[r, :5 * counts[r]] for r in range(len(counts))]).contiguous()
    nrec = records.numel() // 5

    # 2. Newtonian pull of every massive body, and the prefix sums
    head_sums, slice_sums = z(10), z(10)
    ctx.shard_gr_newton(view, 0, K, id_head, records, nrec, head_sums, mode)
    ctx.shard_gr_newton(view, K, n, id_slice, records, nrec, slice_sums, mode)
    all_sums, _ = gather(slice_sums)
    before_slice = head_sums + all_sums[:rank].sum(dim=0)

    # 3. post-Newtonian terms in the Jacobi frames, back-reactions
    head_push, slice_push = z(3), z(3)
    ctx.shard_gr_forces(view, c, max_iterations, 0, K, None, head_push, mode)
    ctx.shard_gr_forces(view, c, max_iterations, K, n, before_slice, slice_push, mode)
    all_push, _ = gather(slice_push)

    # 4. back to inertial accelerations
    ctx.shard_gr_apply(view, 0, K, all_push.sum(dim=0))
    ctx.shard_gr_apply(view, K, n, all_push[rank + 1:].sum(dim=0))


class HipGrFullEngine:
    """gr_full phases on one GPU through librebxamd."""

    def __init__(self, ctx, particles, mode=0):
        self.ctx, self.dp, self.mode = ctx, particles, mode
        self.view = particles.view()
        self.device = particles.cols["x"].device
        self.n = particles.n

    def zeros(self, *shape):
        return torch.zeros(*shape, dtype=torch.float64, device=self.device)

    def flag(self):
        return torch.zeros(1, dtype=torch.int32, device=self.device)

    def newton(self, ignore10, an, a0):
        self.ctx.grfull_newton(self.view, ignore10, an, a0)

    def pot(self, c, r0, r1, pot):
        self.ctx.grfull_pot(self.view, c, r0, r1, pot, self.mode)

    def const(self, c, r0, r1, pot, ac):
        self.ctx.grfull_const(self.view, c, r0, r1, pot, ac, self.mode)

    def sweep(self, c, r0, r1, an, ao, ac, out, flag):
        self.ctx.grfull_sweep(self.view, c, r0, r1, an, ao, ac, out, flag, self.mode)

    def pot_sweep0(self, c, r0, r1, an, pot):
        """The potential sums and, in the same pass over the pairs, the pair sums of the first sweep."""
        self.ctx.grfull_pot_sweep0(self.view, c, r0, r1, an, pot, self.mode)

    def sweep0_finish(self, c, r0, r1, an, ao, ac, out, flag):
        self.ctx.grfull_sweep0_finish(self.view, c, r0, r1, an, ao, ac, out, flag, self.mode)

    def finish(self, a):
        self.ctx.grfull_finish(self.view, a)

    def pack(self, out, r0, per, flag, send):
        """send <- this rank's rows of the new iterate and its vote (one message); the flag is cleared."""
        self.ctx.grfull_pack(out, r0, per, self.n, flag, send)

    def unpack(self, recv, world, per, out, flag, stop, vote):
        """All ranks' messages -> the full columns; the votes -> the stop word and `vote` (recv None: the vote of `flag` alone)."""
        self.ctx.grfull_unpack(recv, world, per, out, flag, stop, vote)

    def set_stop(self, word):
        """word: int32 device tensor; while set, sweeps queued behind a non-zero word return at once."""
        self.ctx.grfull_stop_word(word)


def grfull_sharded(engine, c, gravity_ignore_10=0, group=None):
    """One gr_full evaluation (reference src/gr_full.c:61-231) with rows sharded over the
    ranks of `group`.  Adds the result to the engine's particle accelerations on every rank
    and returns the number of sweeps run.  With no process group it runs all rows locally; with
    one, the collectives run whatever its size (a one-rank group exercises the same calls).

    Communication per evaluation: one all-gather of the potential sums (8 B per body) and ONE
    all-gather per sweep that carries the rank's rows of the three new acceleration columns and
    its "not yet converged" vote in a single message (24 B per body + 8 B): xGMI transfers of this
    size are latency-bound, so the number of collectives is what matters."""
    import torch.distributed as dist
    use_dist = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if use_dist else 1
    rank = dist.get_rank(group) if use_dist else 0
    n = engine.n
    r0, r1, per = row_shard(n, world, rank)
    npad = per * world

    # Workspaces are kept by the engine from one evaluation to the next (a dozen allocations and clears per evaluation
    # are a measurable share of a rank's 2-3 ms at 8 GPUs).  Rows behind n (padding up to a multiple of the world size) are
    # zero from the start and are only ever overwritten with other ranks' padding.
    cache = engine.__dict__.setdefault("_grfull_workspace", {})
    ws = cache.get((npad, per, world))
    if ws is None:
        ws = {"an": [engine.zeros(npad) for _ in range(3)], "ac": [engine.zeros(npad) for _ in range(3)],
              "ao3": engine.zeros(3, npad), "out3": engine.zeros(3, npad),   # rows of one tensor, so that one copy packs them
              "pot": engine.zeros(npad), "flag": engine.flag(),
              "send": engine.zeros(3 * per + 1), "recv_flat": engine.zeros(world * (3 * per + 1))}
        cache.clear()
        cache[(npad, per, world)] = ws
    an, ac, ao3, out3, pot, flag = ws["an"], ws["ac"], ws["ao3"], ws["out3"], ws["pot"], ws["flag"]
    if use_dist:
        send, recv_flat = ws["send"], ws["recv_flat"]
        recv = recv_flat.view(world, 3 * per + 1)
    engine.newton(gravity_ignore_10, an, [ao3[0], ao3[1], ao3[2]])   # replicated: every rank holds the full incoming field
    # (an engine that can lets the first sweep -- a_old = 0: no potentials, no a_const needed yet -- share the pass of the
    # potential sums; its pair sums wait in the engine until a_const exists)
    shared_pass = hasattr(engine, "pot_sweep0")
    if shared_pass:
        engine.pot_sweep0(c, r0, r1, an, pot)
    else:
        engine.pot(c, r0, r1, pot)
    if use_dist:
        _all_gather_into(pot, pot[rank * per:(rank + 1) * per].clone(), group)
    engine.const(c, r0, r1, pot, ac)
    bufs = [ao3, out3]                              # sweep k reads the iterate in bufs[k % 2] and writes bufs[(k + 1) % 2]

    def one_sweep(k, stop=None, vote=None):
        """Queues sweep k and its exchange; returns the device tensor that is non-zero while any rank's rows fail
        the stopping rule.  stop, vote (engines with pack/unpack kernels): the exchange is two small kernels around the
        all-gather, which also turn the votes into the stop word of the queued sweeps and into `vote`."""
        src, dst = bufs[k % 2], bufs[(k + 1) % 2]
        if stop is None:
            flag.zero_()
        run = engine.sweep0_finish if (k == 0 and shared_pass) else engine.sweep
        run(c, r0, r1, an, [src[0], src[1], src[2]], ac, [dst[0], dst[1], dst[2]], flag)
        if stop is not None:
            if use_dist:
                engine.pack([dst[0], dst[1], dst[2]], r0, per, flag, send)
                _all_gather_into(recv_flat, send, group)
                engine.unpack(recv_flat, world, per, [dst[0], dst[1], dst[2]], None, stop, vote)
            else:
                engine.unpack(None, 1, per, [dst[0], dst[1], dst[2]], flag, stop, vote)
            return vote
        if not use_dist:
            return flag
        send[:3 * per].view(3, per).copy_(dst.view(3, world, per)[:, rank, :])
        send[3 * per:].copy_(flag)
        _all_gather_into(recv_flat, send, group)
        dst.view(3, world, per).copy_(recv[:, :3 * per].view(world, 3, per).permute(1, 0, 2))
        return recv[:, 3 * per].max()

    sweeps, converged = 10, False
    if hasattr(engine, "set_stop") and pot.is_cuda:
        # Sweeps are queued one ahead of the host: the vote of sweep k is turned into a stop word on the device
        # (the kernels of a sweep queued behind a set word return at once, writing nothing) and copied to pinned
        # memory without waiting; the host looks at the vote of sweep k - 1 after it has queued sweep k.
        if "votes" not in ws:
            ws["votes"], ws["stop"], ws["vote"] = torch.ones(10, dtype=torch.int32).pin_memory(), engine.flag(), engine.flag()
        votes, stop, vote_dev = ws["votes"], ws["stop"], ws["vote"]
        votes.fill_(1)
        stop.zero_()
        events = []
        fused = hasattr(engine, "unpack")
        if fused:
            flag.zero_()                            # (the exchange kernels clear it after every sweep)
        engine.set_stop(stop)
        try:
            for k in range(10):                     # reference loop bound (gr_full.c:173)
                if fused:
                    votes[k:k + 1].copy_(one_sweep(k, stop, vote_dev), non_blocking=True)
                else:
                    vote = one_sweep(k)
                    unconv = (vote != 0).to(torch.int32).reshape(1)
                    torch.maximum(stop, 1 - unconv, out=stop)
                    votes[k:k + 1].copy_(unconv, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                events.append(ev)
                if k >= 1:
                    events[k - 1].synchronize()
                    if int(votes[k - 1]) == 0:      # sweep k - 1 met the rule: k sweeps ran, sweep k was skipped on the device
                        sweeps, converged = k, True
                        break
            if not converged:
                events[-1].synchronize()
                converged = int(votes[9]) == 0
        finally:
            engine.set_stop(None)
    else:
        for k in range(10):
            if float(one_sweep(k)) == 0.0:
                sweeps, converged = k + 1, True
                break
    if not converged:
        import warnings
        warnings.warn("10 loops in rebx_gr_full did not converge.")      # the reference's message (gr_full.c:221-223)
    newest = bufs[sweeps % 2]
    engine.finish([newest[0], newest[1], newest[2]])
    return sweeps
