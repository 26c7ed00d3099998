# per-kernel average durations of the decode step under the given environment (eager launches, rocprofv3 kernel trace):
#   bash tools/kernel_times.sh llama2_7b L2_TUNE_ROT=0
export L2_TEST_HOOKS=1   # the development switches below only exist behind this gate
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cfg=$1; shift
rm -rf gpurun_out/kt
( export L2_USE_GRAPH=0 "$@"; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -o p -- python3 bench.py --config $cfg --no-cpu-baseline --no-extra --no-dropin --no-pmc --steps 64 --warmup 8 > /dev/null 2>&1 )
echo "== $cfg $*"
python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/kt/p_kernel_stats.csv")):
    n = r["Name"]
    if "l2k" in n and "synth" not in n: print("  %-60s %6s %9.2f us" % (n.replace("l2k::", "").split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
