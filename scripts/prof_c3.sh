#!/bin/bash
# Kernel-level breakdown of the all-massive stack (BASELINE config 3), PARTICLE and Jacobi coordinates (GPU box, repository root).
set -e
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_c3" -o c3 -- python3 "$ROOT/scripts/bench_c3.py" ${1:-1e7} 10 > "$OUT/prof_c3.log" 2>&1
grep "^C3" "$OUT/prof_c3.log"
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/prof_c3/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-90s calls %5s avg %10.1f ns  %5s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
PY
