#!/usr/bin/env python3
"""Per-kernel averages of SQ counters from rocprofv3 --pmc passes (one directory per pass).
usage: python profiles/sq_counters.py <kernel substring> <dir> [<dir> ...]"""
import collections, csv, glob, os, sys
key = sys.argv[1]
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if key in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in sorted(acc.items()):
            v = v[3:] if len(v) > 6 else v      # steady-state launches
            print("%-28s launches %3d  mean %16.1f" % (c, len(v), sum(v) / len(v)))
