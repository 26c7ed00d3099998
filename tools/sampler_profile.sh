# per-kernel times of the device sampler (eager launches so that every kernel is a row)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/sp
L2_USE_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sp -o p -- python3 tools/sampler_bench.py ${1:-stories110M} 48 > /dev/null 2>&1
python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/sp/p_kernel_stats.csv")):
    n = r["Name"]
    if "phase" in n or "synth" in n: continue
    print("  %-80s %6s %9.2f us" % (n.split("(")[0][:80], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
python tools/sampler_bench.py ${1:-stories110M}
