#!/usr/bin/env python3
"""Condenses the raw rocprofv3 outputs of scripts/collect_profiles.sh (gpurun_out/prof_*) into
the files kept under profiles/:
  r01_bench_kernel_stats.csv  per-kernel calls / total / average / min / max of `bench.py --no-cpu`
  traffic.json                HBM bytes per launch of the streaming kernel from the PMC passes
  r01_bench.json              the JSON line that profiled bench run printed
HBM bytes follow the guide's rule for gfx950: FETCH_SIZE (KB) is doubled (wide coalesced loads are
tallied at half their size), WRITE_SIZE (KB) is taken as is; the two come from separate runs."""
import csv
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")


def short(name):
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void ", "")


def kernel_stats():
    rows = list(csv.DictReader(open(os.path.join(OUT, "prof_trace", "bench_kernel_stats.csv"))))
    with open(os.path.join(PROF, "r01_bench_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            if float(r["Percentage"]) < 0.001:
                continue
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])


def counter_mean(path, counter, kernel_substr):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if r["Counter_Name"] == counter and kernel_substr in r["Kernel_Name"]]
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
    json.dump(out, open(os.path.join(PROF, "traffic.json"), "w"), indent=1)
    return out


def bench_line():
    line = [l for l in open(os.path.join(OUT, "prof_trace.log")) if l.startswith('{"metric"')][-1]
    json.dump(json.loads(line), open(os.path.join(PROF, "r01_bench.json"), "w"), indent=1)


if __name__ == "__main__":
    os.makedirs(PROF, exist_ok=True)
    kernel_stats()
    t = traffic()
    bench_line()
    print("traffic %.4g B per launch vs %.4g algorithmic" % (t["hbm_bytes_per_launch"], t["algorithmic_bytes_per_launch"]))
