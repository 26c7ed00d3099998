#!/bin/bash
# Per-kernel durations of the device sampler's launches (rocprofv3 kernel trace of tools/sampler_bench.py, eager launches).
# usage: tools/sampler_kernel_stats.sh <config> <out.txt>
cfg=${1:-stories110M}; out=${2:-gpurun_out/sampler_kernels.txt}
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sk && L2_USE_GRAPH=0 rocprofv3 --kernel-trace --stats -d /tmp/sk -o sk --output-format csv -- python3 $root/tools/sampler_bench.py $cfg 64 > /tmp/sk.log 2>&1
f=$(find /tmp/sk -name '*kernel_stats.csv' | head -1)
python3 - "$f" > $root/$out <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if "l2s::" in n or "argmax" in n:
        print("  %-60s %6s calls  %8.2f us" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
cat /tmp/sk.log | tail -2 >> $root/$out
