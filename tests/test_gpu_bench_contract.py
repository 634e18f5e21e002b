"""bench.py prints ONE JSON line with the fields its contract names (a short run on the GPU)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--n-per-gpu", "200000", "--cpu-reps", "1",
                        "--grfull-n", "2048", "--c5-n", "50000", "--c5-steps", "20", "--c3-n", "100000", "--c3-big-n", "200000", "--ooc-n", "300000"],
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak" and out["vs_baseline"] is None
    assert out["dtype"] == "f64" and out["data"] == "synthetic" and "workload" in out["config"] and out["higher_is_better"] is True
    assert out["value"] > 0 and abs(out["value"] - 200000 * 3 / (out["ms_per_step"] * 3e-3)) <= 1e-5 * out["value"]      # whole-job units / time
    roof = out["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    cpu = out["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["unit"] == "evals/s" and cpu["sample"]
    assert out["resident_run"]["state_finite"] and out["gr_full"]["sweeps"] >= 1 and "jacobi" in out["massive_stack"]
    ooc = out["roofline_out_of_cache"]
    assert ooc["bound"] == "hbm" and ooc["particles"] == 300000 and abs(ooc["frac"] - ooc["achieved"] / ooc["peak"]) < 1e-12
    full = out["massive_stack_full_size"]
    assert full["n_total"] == 200000 and full["scaling"] == "strong" and full["particle"]["ms_per_eval"] > 0 and full["jacobi"]["ms_per_eval"] > 0
    # every part of the metric carries its CPU figure, and numbers that come from committed profiles say so
    for section in (out["gr_full"], out["massive_stack"]):
        cpu = section["cpu_baseline"]
        assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["sample"], cpu
    assert out["gr_full"]["cpu_baseline"]["unit"] == "pair-interactions/s"
    if roof["traffic"] is not None:
        assert "profiles/traffic.json" in roof["traffic_source"] and "not measured in this run" in roof["traffic_source"]
    rr = out["resident_run"]
    if rr["roofline"] is not None:
        assert "not measured in this run" in rr["roofline"]["instructions_source"]
    check = rr["sampled_grains_vs_checker"]
    assert check["grains"] == 256 and check["steps"] == 20 and check["within_bound"] and check["max_deviation"] <= 2e-10


def test_bench_runs_as_two_ranks_with_every_sharded_section():
    """The launch line the scaling runs use (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`), with two
    ranks sharing this GPU and the collectives over gloo (RCCL wants one GPU per rank): the three sharded sections -- the
    headline stack (weak), gr_full (rows over ranks) and the all-massive stack at full size (grains over ranks, strong) --
    all run, rank 0 prints the one line, and it carries the job's size and each section's sharding."""
    import math
    env = dict(os.environ, REBXAMD_DIST_BACKEND="gloo", OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29617", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--n-per-gpu", "200000", "--grfull-n", "2048", "--c5-n", "0", "--c3-n", "0", "--c3-big-n", "200001", "--ooc-n", "0"],
                       capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and "cpu_baseline" not in out
    assert abs(out["value"] - 2 * 200000 * 3 / (out["ms_per_step"] * 3e-3)) <= 1e-5 * out["value"]                        # both ranks' grains
    assert "no communication" in out["config"]["sharding"]
    gr = out["gr_full"]
    assert gr["n_gpus"] == 2 and "rows i over ranks" in gr["sharding"] and 1 <= gr["sweeps"] <= 10
    assert math.isfinite(gr["ms_per_eval"]) and gr["ms_per_eval"] > 0 and gr["roofline"]["peak"] == pytest.approx(2 * 78.6)
    full = out["massive_stack_full_size"]
    assert full["n_gpus"] == 2 and full["n_total"] == 200001 and full["scaling"] == "strong" and "over ranks" in full["sharding"]
    for label in ("particle", "jacobi"):
        assert math.isfinite(full[label]["ms_per_eval"]) and full[label]["ms_per_eval"] > 0 and full[label]["roofline"]["peak"] == pytest.approx(2 * 8000.0)
