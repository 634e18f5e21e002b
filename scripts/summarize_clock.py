"""Condenses the counter pass of scripts/pmc_clock.sh into gpurun_out/clock_valu.json: per kernel the mean duration, the effective
clock (GRBM_GUI_ACTIVE / 8 XCDs / duration) and the VALU instructions per wavefront.  Dispatches that returned at once (gr_full
sweeps queued past convergence) are left out: those shorter than a fifth of the kernel's longest dispatch.
usage: python scripts/summarize_clock.py [gpurun_out]"""
import collections
import csv
import glob
import os
import sys

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
cc = glob.glob(out + "/prof_clock/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(cc)))
longest = collections.defaultdict(float)
for r in rows:
    longest[r["Kernel_Name"]] = max(longest[r["Kernel_Name"]], float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    if ("grfull_fast" in k or "stream_kernel" in k) and dur >= 0.2 * longest[k]:
        acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            acc[k[:60]]["dur_ns"].append(dur)
import json
summary = {"source": "rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --kernel-trace on "
                     "`bench.py --steps 3 --warmup 1 --no-cpu --c5-n 0 --c3-n 0 --c3-big-n 0 --ooc-n 0`; clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration, for dispatches of at least a millisecond", "kernels": {}}
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    # the clock is derived only for dispatches of a millisecond or more: on a 150 us kernel the counter's window and the
    # kernel's timestamps do not line up well enough (it read 2.78 GHz on a chip whose maximum is 2.4)
    long_enough = "dur_ns" in m and m["dur_ns"] >= 1e6
    summary["kernels"][k] = {"dispatches": len(d["SQ_WAVES"]), "duration_us": m.get("dur_ns", 0) / 1e3,
                             "effective_clock_GHz": (m["GRBM_GUI_ACTIVE"] / 8 / m["dur_ns"]) if long_enough else None,
                             "waves": m["SQ_WAVES"], "valu_instructions": m["SQ_INSTS_VALU"], "valu_per_wave": m["SQ_INSTS_VALU"] / m["SQ_WAVES"]}
    line = "%-62s n=%d" % (k, len(d["SQ_WAVES"]))
    if "dur_ns" in m:
        line += "  dur %.1f us" % (m["dur_ns"] / 1e3) + ("  clock %.2f GHz" % (m["GRBM_GUI_ACTIVE"] / 8 / m["dur_ns"]) if long_enough else "  (too short for a clock figure)")
    line += "  VALU/wave %.0f" % (m["SQ_INSTS_VALU"] / m["SQ_WAVES"])
    print(line)
json.dump(summary, open(out + "/clock_valu.json", "w"), indent=1)
