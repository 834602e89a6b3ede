#!/usr/bin/env python3
"""Per-kernel statistics of a rocprofv3 --kernel-trace run of bench.py over its TIMED region only (the last `steps` of
the warmup + steps launches of the hot kernel), plus the idle time between the kernels of a step.
usage: python profiles/timed_region.py <rocprof output dir> [steps=100] [out.csv]"""
import collections, csv, glob, os, sys

HOT = "k_narrow_team"


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    cut = name.find("(")
    return name[:cut] if cut > 0 else name


def main():
    d = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    f = max(glob.glob(os.path.join(d, "*", "*kernel_trace.csv")), key=os.path.getmtime)
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(f))]
    rows.sort()
    hot = [k for k, r in enumerate(rows) if r[2] == HOT]
    # the timed region: from the first kernel after the hot launch `steps + 1` from the end ... to the last step's end
    first_hot = hot[-steps]
    # walk back to the start of that step: the kernel after the previous step's k_publish
    k0 = first_hot
    while k0 > 0 and rows[k0 - 1][2] != "k_publish":
        k0 -= 1
    k1 = hot[-1]
    while k1 + 1 < len(rows) and rows[k1][2] != "k_publish":
        k1 += 1
    reg = rows[k0:k1 + 1]
    acc = collections.defaultdict(list)
    for s, e, n in reg:
        acc[n].append(e - s)
    busy = sum(e - s for s, e, n in reg)
    span = reg[-1][1] - reg[0][0]
    # gaps inside a step (between consecutive kernels, not across the host round trip at a step's end)
    gaps_in = gaps_between = 0
    for a, b in zip(reg, reg[1:]):
        g = max(0, b[0] - a[1])
        if a[2] == "k_publish": gaps_between += g
        else: gaps_in += g
    out = sys.argv[3] if len(sys.argv) > 3 else None
    lines = ['"Name","CallsPerStep","AverageNs","MinNs","MaxNs","NsPerStep"']
    tab = sorted(((n, len(v) / steps, sum(v) / len(v), min(v), max(v), sum(v) / steps) for n, v in acc.items()), key=lambda r: -r[5])
    for r in tab:
        lines.append('"%s",%.2f,%.1f,%d,%d,%.1f' % r)
    lines.append('"(sum over the kernels of a step)",%.2f,,,,%.1f' % (len(reg) / steps, busy / steps))
    lines.append('"(idle between the kernels of a step)",,,,,%.1f' % (gaps_in / steps))
    lines.append('"(idle between two steps: read-back, host, first launch)",,,,,%.1f' % (gaps_between / steps))
    lines.append('"(first kernel start to last kernel end, per step)",,,,,%.1f' % (span / steps))
    txt = "\n".join(lines) + "\n"
    if out: open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
