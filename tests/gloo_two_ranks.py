"""Helper of test_gpu_parity.py (run once per rank as a child process): the row-sharded gr_full
driver and the particle-sharded driver with two ranks that share cuda:0 over the gloo backend
(collectives staged through the host; RCCL refuses two ranks on one device).
usage: python gloo_two_ranks.py RANK WORLD PORT OUT.npz"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from reboundx_amd import synthetic  # noqa: E402
from reboundx_amd.device import Context, DeviceParticles  # noqa: E402
from reboundx_amd.dist import HipGrFullEngine, apply_forces_sharded, grfull_sharded, particle_shard  # noqa: E402
from reboundx_amd.program import build_ops  # noqa: E402


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    ctx = Context(0)
    # gr_full, rows sharded
    cl = synthetic.cluster(1500)
    for k, s in (("ax", "x"), ("ay", "y"), ("az", "z")):
        cl[k] = -cl[s]
    dp = DeviceParticles.from_host(cl, n_head=1500, tail_massless=False)
    sweeps = grfull_sharded(HipGrFullEngine(ctx, dp, 1), 1e2, 0)
    acc = np.array(dp.acc_host())
    # per-particle forces on [head | slice] with massive slices (pushes all-reduced, prefix sums all-gathered)
    n, h = 20001, 2
    cols, pp, _ = synthetic.disk(n, h - 1, dust_mass=1e-12)
    pp["primary"] = np.zeros(n, dtype=np.uint8); pp["primary"][0] = 1
    lo, hi = particle_shard(n, h, world, rank)
    idx = np.concatenate([np.arange(h), np.arange(lo, hi)])
    host = {k: v[idx] for k, v in cols.items()}
    for k in ("beta", "R_tides", "k1", "tau_a", "tau_e", "tau_inc"):
        host[k] = pp[k][idx]
    stack = [("tides_precession", {}), ("modify_orbits_forces", {}), ("gr_potential", {"c": synthetic.C_DEFAULT}), ("gr", {"c": synthetic.C_DEFAULT})]
    dq = DeviceParticles.from_host(host, n_head=h, tail_massless=False)
    apply_forces_sharded(ctx, dq, build_ops(stack, pp), 1)
    np.savez(out, sweeps=sweeps, acc=acc, idx=idx, slice_acc=np.array(dq.acc_host()))
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
