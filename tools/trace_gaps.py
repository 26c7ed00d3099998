"""Summarise a rocprofv3 kernel-trace CSV: per-kernel mean duration and the mean idle gap in front of
each kernel inside a token (graph replay), so launch-bound phases show up as gaps, not kernel time."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
prev_end = None
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void l2k::", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[name].append(e - s)
    if prev_end is not None and 0 <= s - prev_end < 50000:
        gap[name].append(s - prev_end)
    prev_end = e
print("%-46s %7s %9s %9s" % ("kernel", "calls", "dur_us", "gap_us"))
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    g = gap.get(k, [0])
    print("%-46s %7d %9.2f %9.2f" % (k[:46], len(dur[k]), sum(dur[k]) / len(dur[k]) / 1e3, sum(g) / max(len(g), 1) / 1e3))
