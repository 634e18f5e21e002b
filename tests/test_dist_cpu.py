"""The multi-GPU drivers' sharding and exchange logic, exercised with two gloo ranks on CPU.
The compute engine here is the CPU checker (test infrastructure); on GPUs the same driver
code runs on the HIP engine (tests/test_gpu_parity.py covers that engine's row-range calls)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle  # noqa: E402
from reboundx_amd import synthetic  # noqa: E402
from reboundx_amd.dist import grfull_sharded, particle_shard, row_shard  # noqa: E402


class OracleGrFullEngine:
    def __init__(self, cols, G=1.0):
        self.sim = oracle.Sim(cols, G)
        self.n = self.sim.N

    def zeros(self, *shape):
        return torch.zeros(*shape, dtype=torch.float64)

    def flag(self):
        return torch.zeros(1, dtype=torch.int32)

    @staticmethod
    def _p(t):
        import ctypes
        return ctypes.c_void_p(t.data_ptr())

    def newton(self, ig, an, a0):
        oracle.lib.orc_gr_full_newton(__import__("ctypes").byref(self.sim.s), ig, self._p(an[0]), self._p(an[1]), self._p(an[2]))
        for t in a0:
            t.zero_()

    def pot(self, c, r0, r1, pot):
        import ctypes
        oracle.lib.orc_gr_full_pot(ctypes.byref(self.sim.s), ctypes.c_double(c), r0, r1, self._p(pot))

    def const(self, c, r0, r1, pot, ac):
        import ctypes
        oracle.lib.orc_gr_full_const(ctypes.byref(self.sim.s), ctypes.c_double(c), r0, r1, self._p(pot), *[self._p(t) for t in ac])

    def sweep(self, c, r0, r1, an, ao, ac, out, flag):
        import ctypes
        oracle.lib.orc_gr_full_sweep.restype = ctypes.c_int
        f = oracle.lib.orc_gr_full_sweep(ctypes.byref(self.sim.s), ctypes.c_double(c), r0, r1, *[self._p(t) for t in an + ao + ac + out])
        flag[0] = max(int(flag[0]), f)

    def finish(self, a):
        n = self.n
        for k, t in zip(("ax", "ay", "az"), a):
            self.sim.cols[k] += t.numpy()[:n]


def _cluster(n):
    cl = synthetic.cluster(n)
    s = oracle.Sim(cl)
    oracle.newtonian_gravity(s)
    cl["ax"], cl["ay"], cl["az"] = [a.copy() for a in s.acc]
    return cl


def _worker(rank, world, port, n, ig, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng = OracleGrFullEngine(_cluster(n))
        sweeps = grfull_sharded(eng, 1e2, ig)
        q.put((rank, sweeps, np.array(eng.sim.acc)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,ig", [(37, 0), (64, 1)])
def test_gr_full_row_sharding_two_ranks_equals_single_process(n, ig):
    ref = oracle.Sim(_cluster(n))
    sweeps_ref = oracle.gr_full(ref, 1e2, ig)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n) % 1000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, ig, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, sweeps, acc in results:
        assert sweeps == sweeps_ref
        assert np.array_equal(acc, np.array(ref.acc)), rank      # same bits on every rank


def test_gr_full_driver_without_process_group_is_the_plain_evaluation():
    n = 23
    eng = OracleGrFullEngine(_cluster(n))
    ref = oracle.Sim(_cluster(n))
    assert grfull_sharded(eng, 1e2, 0) == oracle.gr_full(ref, 1e2, 0)
    assert np.array_equal(np.array(eng.sim.acc), np.array(ref.acc))


def test_shards_tile_the_index_space():
    for n, w in ((10, 3), (65536, 8), (7, 8), (1, 2)):
        rows = [row_shard(n, w, r) for r in range(w)]
        assert rows[0][0] == 0 and rows[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
    for n, h, w in ((1000, 3, 4), (10_000_001, 1, 8), (5, 5, 2)):
        parts = [particle_shard(n, h, w, r) for r in range(w)]
        assert parts[0][0] == h and parts[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))


def test_particle_shards_reproduce_the_unsharded_update():
    """Behind the head every particle's update depends on the head and on itself only, so
    evaluating [head | shard] per rank and concatenating equals evaluating everything."""
    n, h, w = 3000, 3, 4
    cols, pp, _ = synthetic.disk(n, h - 1)
    synthetic.add_newtonian(cols, n_head=h)
    stack = [("radiation_forces", {"c": synthetic.C_DEFAULT}), ("gravitational_harmonics", {}), ("gr", {"c": synthetic.C_DEFAULT})]
    full = oracle.Sim(cols)
    oracle.apply_forces(full, stack, pp)
    pieces = []
    for r in range(w):
        lo, hi = particle_shard(n, h, w, r)
        idx = np.concatenate([np.arange(h), np.arange(lo, hi)])
        sub = oracle.Sim({k: v[idx] for k, v in cols.items()})
        oracle.apply_forces(sub, stack, {k: v[idx] for k, v in pp.items()})
        acc = np.array(sub.acc)
        if r == 0:
            pieces.append(acc[:, :h])
        else:
            assert np.array_equal(acc[:, :h], pieces[0])    # the replicated head agrees on every rank
        pieces.append(acc[:, h:])
    assert np.array_equal(np.concatenate(pieces, axis=1), np.array(full.acc))


class NumpyShardEngine:
    """The rebxamd_shard_* calls of the sharded gr driver restated with sequential numpy loops
    (test infrastructure: lets the driver's exchange logic run over gloo without a GPU).
    Post-Newtonian term: reference src/gr.c:110-156."""

    class Particles:
        def __init__(self, cols, n_head):
            self.c = {k: np.array(v, dtype=np.float64) for k, v in cols.items()}
            self.n_head, self.n = n_head, len(cols["x"])
            self.cols = {"x": torch.zeros(1, dtype=torch.float64)}

        def view(self):
            return self

    def __init__(self, G=1.0):
        self.G, self.na, self.aj, self.w = G, {}, {}, {}

    def shard_massive(self, p, lo, hi, id0, records, count):
        k = 0
        for i in range(lo, hi):
            if p.c["m"][i] != 0.0:
                records[5 * k:5 * k + 5] = torch.tensor([p.c["x"][i], p.c["y"][i], p.c["z"][i], self.G * p.c["m"][i], float(id0 + i - lo)], dtype=torch.float64)
                k += 1
        count[0] = float(k)

    def shard_gr_newton(self, p, lo, hi, id0, records, nrec, sums10, mode=0):
        rec = records.numpy().reshape(-1, 5)[:nrec]
        na = np.zeros((hi - lo, 3))
        s = np.zeros(10)
        for i in range(lo, hi):
            r = np.array([p.c["x"][i], p.c["y"][i], p.c["z"][i]])
            for q in rec:
                if q[4] == id0 + i - lo:
                    continue
                d = q[:3] - r
                na[i - lo] += q[3] * d / np.dot(d, d) ** 1.5
            m = p.c["m"][i]
            s += m * np.concatenate([[1.0], r, [p.c["vx"][i], p.c["vy"][i], p.c["vz"][i]], na[i - lo]])
        self.na[lo] = na
        sums10.copy_(torch.from_numpy(s))

    def shard_gr_forces(self, p, c, max_iterations, lo, hi, before10, push3, mode=0):
        S = before10.numpy().copy() if before10 is not None else np.zeros(10)
        mu, C2 = self.G * p.c["m"][0], c * c
        aj, w = np.zeros((hi - lo, 3)), np.zeros((hi - lo, 3))
        for i in range(lo, hi):
            m = p.c["m"][i]
            st = np.array([p.c[k][i] for k in ("x", "y", "z", "vx", "vy", "vz")] + list(self.na[lo][i - lo]))
            if not (lo == 0 and i == 0):
                j = st - S[1:] / S[0]
                x, v, a = j[:3], j[3:6], j[6:]
                ri = np.sqrt(np.dot(x, x))
                vi = v.copy()
                A = (0.5 * np.dot(vi, vi) + 3.0 * mu / ri) / C2
                for _ in range(max_iterations):
                    old = vi
                    vi = v / (1.0 - A)
                    A = (0.5 * np.dot(vi, vi) + 3.0 * mu / ri) / C2
                    if np.dot(vi - old, vi - old) / np.dot(vi, vi) < np.finfo(float).eps ** 2:
                        break
                B = (mu / ri - 1.5 * np.dot(vi, vi)) * mu / ri ** 3 / C2
                D = (np.dot(vi, a + B * x) - 3.0 * mu / ri ** 3 * np.dot(x, v)) / C2
                aj[i - lo] = B * (1.0 - A) * x - A * a - D * vi
                w[i - lo] = m / (S[0] + m) * aj[i - lo]
            S += m * np.concatenate([[1.0], st])
        self.aj[lo], self.w[lo] = aj, w
        push3.copy_(torch.from_numpy(w.sum(axis=0)))

    def shard_gr_apply(self, p, lo, hi, after3):
        run = after3.numpy().copy() if after3 is not None else np.zeros(3)
        for i in range(hi - 1, lo - 1, -1):
            run = run + self.w[lo][i - lo]
            for k, name in enumerate(("ax", "ay", "az")):
                p.c[name][i] += self.aj[lo][i - lo][k] - run[k]


def _mixed_mass_system(n, h):
    cols, pp, _ = synthetic.disk(n, h - 1)
    rng = np.random.default_rng(5)
    for i in rng.choice(np.arange(h, n), size=7, replace=False):
        cols["m"][i] = 10.0 ** rng.uniform(-6, -3)
    synthetic.add_newtonian(cols, n_head=h)
    return cols, pp


def _gr_worker(rank, world, port, n, h, q):
    from reboundx_amd.dist import gr_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cols, _ = _mixed_mass_system(n, h)
        lo, hi = particle_shard(n, h, world, rank)
        idx = np.concatenate([np.arange(h), np.arange(lo, hi)])
        p = NumpyShardEngine.Particles({k: v[idx] for k, v in cols.items()}, h)
        gr_sharded(NumpyShardEngine(), p, synthetic.C_DEFAULT, 10, 1)
        q.put((rank, idx, np.array([p.c["ax"], p.c["ay"], p.c["az"]])))
    finally:
        dist.destroy_process_group()


def test_sharded_gr_driver_two_ranks_with_massive_bodies_in_the_slices():
    """gr_sharded over gloo, uneven slices (the tail length is odd), massive bodies scattered in
    both slices: the assembled accelerations equal the sequential checker's on the whole system."""
    n, h = 48, 2
    cols, pp = _mixed_mass_system(n, h)
    ref = oracle.Sim(cols)
    oracle.apply_forces(ref, [("gr", {"c": synthetic.C_DEFAULT})], pp)
    ref = np.array(ref.acc)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 333) % 1000
    procs = [ctx.Process(target=_gr_worker, args=(r, 2, port, n, h, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got = np.zeros((3, n))
    heads = []
    for rank, idx, acc in results:
        got[:, idx[h:]] = acc[:, h:]
        heads.append(acc[:, :h])
    assert np.array_equal(heads[0], heads[1])
    got[:, :h] = heads[0]
    base = np.array([cols["ax"], cols["ay"], cols["az"]])
    scale = np.abs(ref - base).max(axis=0)[None, :]
    assert (np.abs(got - ref) / np.maximum(scale, 1e-300)).max() <= 1e-10
