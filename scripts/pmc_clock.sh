#!/bin/bash
# Effective clock and VALU issue of the dominant kernels (one PMC pass; GPU box, repository root):
# clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration (valid for dispatches of a millisecond or more).
set -e
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/prof_clock" -o clk -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu --c5-n 0 --c3-n 0 --c3-big-n 0 --ooc-n 0 > "$OUT/prof_clock.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
cc = glob.glob(out + "/prof_clock/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(cc)))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "grfull_fast" in k or "stream_kernel" in k:
        acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "Start_Timestamp" in r and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            acc[k[:60]]["dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
import json
summary = {"source": "rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --kernel-trace on "
                     "`bench.py --steps 3 --warmup 1 --no-cpu --c5-n 0 --c3-n 0 --c3-big-n 0 --ooc-n 0`; clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration", "kernels": {}}
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    summary["kernels"][k] = {"dispatches": len(d["SQ_WAVES"]), "duration_us": m.get("dur_ns", 0) / 1e3,
                             "effective_clock_GHz": (m["GRBM_GUI_ACTIVE"] / 8 / m["dur_ns"]) if "dur_ns" in m else None,
                             "waves": m["SQ_WAVES"], "valu_instructions": m["SQ_INSTS_VALU"], "valu_per_wave": m["SQ_INSTS_VALU"] / m["SQ_WAVES"]}
    line = "%-62s n=%d" % (k, len(d["SQ_WAVES"]))
    if "dur_ns" in m:
        line += "  dur %.1f us  clock %.2f GHz" % (m["dur_ns"] / 1e3, m["GRBM_GUI_ACTIVE"] / 8 / m["dur_ns"])
    line += "  VALU/wave %.0f" % (m["SQ_INSTS_VALU"] / m["SQ_WAVES"])
    print(line)
json.dump(summary, open(out + "/clock_valu.json", "w"), indent=1)
PY
