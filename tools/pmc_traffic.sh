# HBM traffic counters for the weight-streaming kernels of one config: bash tools/pmc_traffic.sh <config>
# (separate --pmc passes, kernel-trace only, as the MI355X guide prescribes)
cfg=${1:-stories110M}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_$cfg
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/pmc_$cfg/$ctr -o p -- python3 tools/pmc_target.py $cfg > /dev/null 2>&1
done
python3 tools/pmc_summary.py $(find gpurun_out/pmc_$cfg -name "*counter_collection.csv") > gpurun_out/pmc_$cfg/summary.json
cat gpurun_out/pmc_$cfg/summary.json | head -60
