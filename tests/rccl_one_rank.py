"""Helper of test_gpu_parity.py (run as a child process): the multi-GPU drivers of
reboundx_amd/dist.py over a one-rank RCCL process group on cuda:0.  A one-rank group goes
through the same torch.distributed calls (all_gather_into_tensor, all_reduce) with the same
buffer shapes as a larger one; results must equal the runs without a process group."""
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
from reboundx_amd.dist import HipGrFullEngine, apply_forces_sharded, grfull_sharded  # noqa: E402
from reboundx_amd.program import build_ops  # noqa: E402


def main():
    torch.cuda.set_device(0)
    ctx = Context(0)
    C = synthetic.C_DEFAULT
    # --- without a process group
    cl = synthetic.cluster(600)
    for k, s in (("ax", "x"), ("ay", "y"), ("az", "z")):
        cl[k] = -cl[s]
    dp = DeviceParticles.from_host(cl, n_head=600, tail_massless=False)
    sweeps0 = grfull_sharded(HipGrFullEngine(ctx, dp, 1), 1e2, 0)
    acc0 = np.array(dp.acc_host())

    n = 30001
    cols, pp, _ = synthetic.disk(n, 1, dust_mass=1e-14)
    pp["primary"] = np.zeros(n, dtype=np.uint8); pp["primary"][0] = 1
    host = dict(cols)
    for k in ("beta", "R_tides", "k1", "tau_a", "tau_e", "tau_inc"):
        host[k] = pp[k]
    stack = [("tides_precession", {}), ("modify_orbits_forces", {}), ("gr_potential", {"c": C}), ("gravitational_harmonics", {})]
    ops = build_ops(stack, {k: v[:1] for k, v in pp.items()})
    dq = DeviceParticles.from_host(host, n_head=1, tail_massless=False)
    apply_forces_sharded(ctx, dq, ops, 1)
    push0 = np.array(dq.acc_host())

    # --- the same over RCCL
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % (29500 + os.getpid() % 2000), rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    dp = DeviceParticles.from_host(cl, n_head=600, tail_massless=False)
    sweeps1 = grfull_sharded(HipGrFullEngine(ctx, dp, 1), 1e2, 0)
    acc1 = np.array(dp.acc_host())
    dq = DeviceParticles.from_host(host, n_head=1, tail_massless=False)
    apply_forces_sharded(ctx, dq, ops, 1)
    push1 = np.array(dq.acc_host())
    t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    dist.destroy_process_group()
    assert sweeps0 == sweeps1, (sweeps0, sweeps1)
    assert np.array_equal(acc0, acc1), "gr_full over RCCL differs"
    assert np.array_equal(push0, push1), "sharded per-particle forces over RCCL differ"
    assert float(t.item()) == 1.5
    ctx.close()
    print("rccl one-rank ok")


if __name__ == "__main__":
    main()
