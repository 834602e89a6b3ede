#!/usr/bin/env python3
"""Turns the rocprofv3 output directories of one measurement session (gpurun_out/, scratch) into the small
summaries committed under profiles/.  Usage:

    python profiles/make_summaries.py r01 gpurun_out/prof22_stats gpurun_out/prof22_fetch gpurun_out/prof22_write

The three directories come from (run `cd /tmp && export TMPDIR=/tmp` first):
    rocprofv3 --kernel-trace --stats --output-format csv -d <stats> -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <fetch> -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <write> -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline
(counters in their own passes, as MI355X_MICROARCH.md prescribes; FETCH_SIZE doubled for gfx950).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

HOT = "k_narrow_team"   # the kernel bench.py's roofline object prices (rounds 1-2: k_narrow_convex)


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    cut = name.find("(")
    return name[:cut] if cut > 0 else name


def counters(d, counter):
    f = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


def durations(d):
    f = glob.glob(os.path.join(d, "*", "*kernel_trace.csv"))[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return acc


def main():
    tag, stats, fetch, write = sys.argv[1:5]
    here = os.path.dirname(os.path.abspath(__file__))
    shutil.copy(glob.glob(os.path.join(stats, "*", "*kernel_stats.csv"))[0], os.path.join(here, tag + "_kernel_stats.csv"))
    # (the statistics over the TIMED region alone, with the idle time between the kernels: profiles/timed_region.py)
    F, Wr, D = counters(fetch, "FETCH_SIZE"), counters(write, "WRITE_SIZE"), durations(fetch)
    rows = []
    for k in F:
        f = sum(F[k]) / len(F[k]); w = sum(Wr[k]) / len(Wr[k]) if k in Wr else 0.0
        rows.append((k, len(F[k]), f, w, sum(D[k]) / len(D[k])))
    rows.sort(key=lambda r: -(2 * r[2] + r[3]))
    with open(os.path.join(here, tag + "_pmc_per_kernel.txt"), "w") as out:
        out.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 12 --warmup 3 "
                  "--no-cpu-baseline, per-launch averages\n")
        out.write("%-44s %6s %14s %14s %12s\n" % ("kernel", "calls", "FETCH_SIZE KB", "WRITE_SIZE KB", "avg ns"))
        for r in rows:
            out.write("%-44s %6d %14.1f %14.1f %12.0f\n" % r)
    # steady-state launches of the hot kernel only (the first two steps of a fresh field send most pairs to the
    # general sweep: every neighbour pair shares an edge exactly)
    f, w, d = F[HOT][3:], Wr[HOT][3:], D[HOT][3:]
    fk, wk = sum(f) / len(f), sum(w) / len(w)
    json.dump({
        "kernel": HOT, "calls": len(f),
        "FETCH_SIZE_KB_per_launch": fk, "WRITE_SIZE_KB_per_launch": wk,
        "hbm_bytes_per_launch_raw": (fk + wk) * 1024, "hbm_bytes_per_launch": (2 * fk + wk) * 1024,
        "note": "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction of "
                "MI355X_MICROARCH.md, HBM section; uncalibrated for this kernel's narrow scattered accesses); separate "
                "--pmc passes for FETCH_SIZE and WRITE_SIZE; launches 4.. of the run (steady state)",
        "avg_ns_under_pmc": sum(d) / len(d)}, open(os.path.join(here, tag + "_pmc_narrow.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
