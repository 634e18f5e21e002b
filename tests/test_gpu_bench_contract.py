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
