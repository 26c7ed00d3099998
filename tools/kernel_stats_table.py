"""Per-kernel table of a rocprofv3 --kernel-trace --stats run: python tools/kernel_stats_table.py <output dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/p_kernel_stats.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if ("l2k" in r["Name"] or "tp_" in r["Name"] or "l2s" in r["Name"]) and "synth" not in r["Name"] and "pack_kernel" not in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows:
    print("  %-70s %6s %9.2f us  %5.1f %%" % (r["Name"].replace("l2k::", "").replace("void ", "").split("(")[0][:70], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
