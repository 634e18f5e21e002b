#!/bin/bash
# Kernel-level breakdown of scripts/bench_next_rows.py (GPU box, repository root).
set -e
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d "$OUT/prof_next" -o nx -- python3 "$ROOT/scripts/bench_next_rows.py" ${1:-1e7} > "$OUT/prof_next.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
for pat in ("kernel_stats", "memory_copy_stats"):
    fs = glob.glob(sys.argv[1] + "/prof_next/**/*%s.csv" % pat, recursive=True)
    if not fs: continue
    for r in list(csv.DictReader(open(fs[0])))[:16]:
        print("%-90s calls %5s avg %10.1f ns  %5s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
PY
