"""Per-evaluation overhead of the sharded gr_full driver (development aid): the same evaluation
through rebxamd_grfull (native loop), through the Python driver without a process group, and
through the Python driver over a one-rank RCCL group (all collectives issued).
usage: python scripts/grfull_overhead.py [n]"""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from reboundx_amd import synthetic  # noqa: E402
from reboundx_amd.device import Context, DeviceParticles  # noqa: E402
from reboundx_amd.dist import HipGrFullEngine, grfull_sharded  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 16384
torch.cuda.set_device(0)
ctx = Context(0)
cl = synthetic.cluster(n)
dp = DeviceParticles.from_host(cl, n_head=n, tail_massless=False)
for k, s in (("ax", "x"), ("ay", "y"), ("az", "z")):
    dp.cols[k].copy_(-dp.cols[s])
eng = HipGrFullEngine(ctx, dp, 0)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print("n = %d" % n)
print("  native loop (rebxamd_grfull)        %8.3f ms" % timed(lambda: ctx.grfull(dp.view(), 1e4, 0, 0)))
print("  Python driver, no process group     %8.3f ms" % timed(lambda: grfull_sharded(eng, 1e4, 0)))
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % (29500 + os.getpid() % 2000), rank=0, world_size=1, device_id=torch.device("cuda", 0))
print("  Python driver, one-rank RCCL group  %8.3f ms" % timed(lambda: grfull_sharded(eng, 1e4, 0)))
dist.destroy_process_group()
