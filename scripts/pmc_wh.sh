#!/bin/bash
# Instruction mix of the pass that carries a Wisdom-Holman step (one PMC pass; GPU box, repository root).
# usage: bash scripts/pmc_wh.sh [grains] [steps of the queued run]
# Dispatches of one kernel are grouped by their instruction count: a pass that carries a batch of steps issues in
# proportion to the steps of its batch.
set -e
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out"
N=${1:-10000000}
STEPS=${2:-40}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d "$OUT/prof_wh" -o wh -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu --grfull-n 0 --c3-n 0 --c3-big-n 0 --ooc-n 0 --c5-n $N --c5-steps $STEPS > "$OUT/prof_wh.log" 2>&1
python3 - "$OUT" <<'P'
import collections, csv, glob, sys
out = sys.argv[1]
cc = glob.glob(out + "/prof_wh/**/*counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    k = r["Kernel_Name"]
    if "stream_kernel" not in k: continue
    d = per[(k, r["Dispatch_Id"])]
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.) + float(r["Counter_Value"])
    d["dur"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for (k, _), d in per.items():
    key = "%s | %d VALU/wave" % (k[-60:], round(d["SQ_INSTS_VALU"] / d["SQ_WAVES"] / 50.) * 50)
    for c, v in d.items(): acc[key][c].append(v)
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print(k, "n=%d dur %.1f us" % (len(d["SQ_WAVES"]), m["dur"] / 1e3))
    print("   waves %.0f  VALU/wave %.0f  SALU/wave %.0f  clock %.2f GHz  VALU-active/busy %.3f  wait_any/wave_cycles %.3f" % (
        m["SQ_WAVES"], m["SQ_INSTS_VALU"] / m["SQ_WAVES"], m["SQ_INSTS_SALU"] / m["SQ_WAVES"], m["GRBM_GUI_ACTIVE"] / 8 / m["dur"],
        m["SQ_ACTIVE_INST_VALU"] / m["SQ_BUSY_CYCLES"], m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]))
    print("   raw", {c: v for c, v in m.items()})
# Summary for bench.py (copied to profiles/<round>_wh_valu.json): the passes that carry a whole batch of steps are the
# dispatches with the most instructions per wavefront of their kernel instance; the first such instance the bench runs is the
# five-effect stack behind the star, the second the radiation stack behind star + planet.
import json, subprocess
BATCH = 32
first, full = {}, collections.defaultdict(list)
for (k, disp), d in per.items():
    if ", 7>" not in k:            # (the last template argument: 7 = the pass that carries a batch of steps)
        continue
    first[k] = min(first.get(k, 1 << 62), int(disp))
    full[k].append((d["SQ_INSTS_VALU"] / d["SQ_WAVES"], d))
best = {}
for k, rows in full.items():
    # whole batches: within 10 % of the instance's largest count (a run's last pass may carry fewer steps); their median
    top = max(v for v, _ in rows)
    whole = sorted((r for r in rows if r[0] >= 0.9 * top), key=lambda r: r[0])
    best[k] = whole[len(whole) // 2] + (len(whole),)
names = sorted(best, key=lambda k: first[k])
summary = {"source": "rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES ... --kernel-trace on `bench.py --steps 2 --warmup 1 --no-cpu --grfull-n 0 --c3-n 0 "
                     "--c3-big-n 0 --ooc-n 0 --c5-n N --c5-steps STEPS` (scripts/pmc_wh.sh); per kernel instance the dispatches that carry a whole batch of %d steps" % BATCH,
           "steps_per_pass": BATCH}
try:
    summary["commit"] = subprocess.check_output(["git", "-C", out + "/..", "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
except Exception:
    pass
for key, k in zip(("five_effects", "star_planet_radiation"), names):
    v, d, count = best[k]
    summary[key] = {"kernel": k[:90], "valu_per_wave_per_pass": v, "valu_per_wave_and_step": v / BATCH, "waves": d["SQ_WAVES"], "duration_us": d["dur"] / 1e3,
                    "dispatches": count}
json.dump(summary, open(out + "/wh_valu.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
P
