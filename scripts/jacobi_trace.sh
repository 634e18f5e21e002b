#!/bin/bash
# Per-kernel times of the all-massive stack (PARTICLE and Jacobi coordinates, 1e7 bodies): rocprofv3 kernel trace of the
# bench's massive_stack section alone.  Run on the GPU box from the repository root; prints the summary's top rows.
set -e
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_jacobi" -o jac -- python3 "$ROOT/bench.py" --n-per-gpu 1000 --steps 1 --warmup 0 --no-cpu --grfull-n 0 --c5-n 0 --c3-big-n 0 --ooc-n 0 > "$OUT/prof_jacobi.log" 2>&1
python3 - "$OUT/prof_jacobi" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print("%-110s calls %5s avg %10.1f us" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
