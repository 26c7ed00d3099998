cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in llama2_7b stories110M; do
rm -rf gpurun_out/fw_$cfg
L2_USE_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fw_$cfg -o p -- python3 bench.py --config $cfg --no-cpu-baseline --no-extra --no-dropin --no-pmc --steps 64 --warmup 8 > /dev/null 2>&1
python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/fw_$cfg/p_kernel_stats.csv")):
    n = r["Name"]
    if "l2k" in n and "synth" not in n: print("  %-60s %6s %9.2f us" % (n.replace("l2k::", "").split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
