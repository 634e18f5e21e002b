#!/usr/bin/env python3
"""Condenses the raw rocprofv3 outputs of scripts/collect_profiles.sh (gpurun_out/prof_*) into
the files kept under profiles/:
  <round>_bench_kernel_stats.csv  per-kernel calls / total / average / min / max of `bench.py --no-cpu --ooc-n 0`
  <round>_out_of_cache_kernel_stats.csv  the same of the out-of-cache section alone (the headline kernel on 4e7 grains)
  traffic.json                    HBM bytes per launch of the streaming kernel from the PMC passes (headline size, and 4e7 grains: out of cache)
  <round>_bench.json              the JSON line that profiled bench run printed
usage: python scripts/summarize_profiles.py [round, default r02]
HBM bytes follow the guide's rule for gfx950: FETCH_SIZE (KB) is doubled (wide coalesced loads are
tallied at half their size), WRITE_SIZE (KB) is taken as is; the two come from separate runs."""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r02"


def short(name):
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void ", "")


def condense(src, dst):
    rows = list(csv.DictReader(open(src)))
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            if float(r["Percentage"]) < 0.001:
                continue
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])


def kernel_stats():
    condense(os.path.join(OUT, "prof_trace", "bench_kernel_stats.csv"), os.path.join(PROF, ROUND + "_bench_kernel_stats.csv"))
    ooc = os.path.join(OUT, "prof_trace_ooc", "ooc_kernel_stats.csv")
    if os.path.exists(ooc):
        condense(ooc, os.path.join(PROF, ROUND + "_out_of_cache_kernel_stats.csv"))


def side_profiles():
    """scripts/prof_c3.sh, prof_resident.sh, pmc_clock.sh: the all-massive stack, the device-resident run, clock and VALU issue."""
    for src, name in (("prof_c3/c3_kernel_stats.csv", "_massive_stack_kernel_stats.csv"), ("prof_resident/res_kernel_stats.csv", "_resident_run_kernel_stats.csv")):
        path = os.path.join(OUT, src)
        if os.path.exists(path):
            condense(path, os.path.join(PROF, ROUND + name))
    clk = os.path.join(OUT, "clock_valu.json")
    if os.path.exists(clk):
        json.dump(json.load(open(clk)), open(os.path.join(PROF, ROUND + "_clock_valu.json"), "w"), indent=1)


def counter_mean(path, counter, kernel_substr, at_least=0.):
    """Mean of a counter over the dispatches of a kernel (those whose value reaches `at_least`: the run that measures the
    4e7-grain kernel also launches it on a handful of grains)."""
    if not os.path.exists(path):
        path = glob.glob(os.path.join(os.path.dirname(path), "**", os.path.basename(path)), recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if r["Counter_Name"] == counter and kernel_substr in r["Kernel_Name"]]
    vals = [v for v in vals if v >= at_least]
    return sum(vals) / len(vals), len(vals)


def traffic():
    fetch, nf = counter_mean(os.path.join(OUT, "prof_fetch", "fetch_counter_collection.csv"), "FETCH_SIZE", "stream_kernel")
    write, nw = counter_mean(os.path.join(OUT, "prof_write", "write_counter_collection.csv"), "WRITE_SIZE", "stream_kernel")
    line = [l for l in open(os.path.join(OUT, "prof_fetch.log")) if l.startswith('{"metric"')][-1]
    algo = json.loads(line)["roofline"]["algorithmic_bytes_per_launch"]
    out = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --steps 5 --warmup 2 --no-cpu "
                  "--grfull-n 0 --c5-n 0 --c3-n 0`, stream_kernel dispatches averaged (%d / %d dispatches)" % (nf, nw),
        "FETCH_SIZE_KB_raw": fetch,
        "WRITE_SIZE_KB_raw": write,
        "correction": "FETCH_SIZE doubled (gfx950 tallies 128-B requests of wide coalesced loads at 64 B); WRITE_SIZE as is",
        "fetch_bytes": 2 * fetch * 1024,
        "write_bytes": write * 1024,
        "hbm_bytes_per_launch": 2 * fetch * 1024 + write * 1024,
        "algorithmic_bytes_per_launch": algo,
    }
    try:
        fo, nfo = counter_mean(os.path.join(OUT, "prof_fetch_ooc", "fetch_counter_collection.csv"), "FETCH_SIZE", "stream_kernel", 1e5)
        wo, nwo = counter_mean(os.path.join(OUT, "prof_write_ooc", "write_counter_collection.csv"), "WRITE_SIZE", "stream_kernel", 1e5)
        out["out_of_cache"] = {
            "source": "the same two passes on `bench.py --n-per-gpu 1000 --steps 1 --warmup 0 --no-cpu --grfull-n 0 --c5-n 0 --c3-n 0 --c3-big-n 0 "
                      "--ooc-n 40000000`, the stream_kernel dispatches over 4e7 grains averaged (%d / %d dispatches)" % (nfo, nwo),
            "particles": 40000000,
            "FETCH_SIZE_KB_raw": fo, "WRITE_SIZE_KB_raw": wo,
            "fetch_bytes": 2 * fo * 1024, "write_bytes": wo * 1024,
            "hbm_bytes_per_launch": 2 * fo * 1024 + wo * 1024,
            "algorithmic_bytes_per_launch": 112 * 40000000,
        }
    except (IndexError, OSError, ZeroDivisionError) as e:
        print("no out-of-cache counter passes found:", e)
    json.dump(out, open(os.path.join(PROF, "traffic.json"), "w"), indent=1)
    return out


def bench_line():
    line = [l for l in open(os.path.join(OUT, "prof_trace.log")) if l.startswith('{"metric"')][-1]
    json.dump(json.loads(line), open(os.path.join(PROF, ROUND + "_bench.json"), "w"), indent=1)


if __name__ == "__main__":
    os.makedirs(PROF, exist_ok=True)
    kernel_stats()
    side_profiles()
    t = traffic()
    bench_line()
    print("traffic %.4g B per launch vs %.4g algorithmic" % (t["hbm_bytes_per_launch"], t["algorithmic_bytes_per_launch"]))
