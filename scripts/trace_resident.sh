#!/bin/bash
# Timeline of the device-resident stepping section (GPU box, repository root): start-to-start spacing of the passes over
# the massless bodies and the duration of every kernel of a step, from a kernel trace.
set -e
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT/prof_trace_res" -o res -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu --grfull-n 0 --c3-n 0 --c3-big-n 0 --ooc-n 0 --c5-steps 400 > "$OUT/prof_trace_res.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/prof_trace_res/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"][:70]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")))
for k, v in by.items():
    if "rebxamd" not in k or len(v) < 50:
        continue
    tail = v[-100:]
    gaps = [b[0] - a[0] for a, b in zip(tail, tail[1:])]
    durs = [e - s for s, e, _ in tail]
    print("%-72s n=%5d  dur %7.1f us  start-to-start %7.1f us  queues %s" % (k, len(v), sum(durs) / len(durs) / 1e3, sum(gaps) / len(gaps) / 1e3, sorted(set(q for _, _, q in tail))))
PY
