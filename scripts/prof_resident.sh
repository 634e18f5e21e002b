#!/bin/bash
# Kernel-level breakdown of the device-resident stepping section (run on the GPU box from the repository root).
set -e
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_resident" -o res -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu --grfull-n 0 --c3-n 0 --c3-big-n 0 --ooc-n 0 --c5-steps 2000 > "$OUT/prof_resident.log" 2>&1
tail -1 "$OUT/prof_resident.log" | cut -c1-300
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/prof_resident/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print("%-70s calls %6s avg %9.1f ns  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
PY
