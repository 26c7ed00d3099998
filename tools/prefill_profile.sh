# per-kernel prefill GEMM times (7B width) for a few launch variants: bash tools/prefill_profile.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {
  tag=$1; shift
  rm -rf gpurun_out/pf_$tag
  env "$@" true
  ( export "$@"; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pf_$tag -o p -- python3 tools/pf_target.py > /dev/null 2>&1 )
  echo "== $tag ($*)"
  python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/pf_$tag/p_kernel_stats.csv")):
    n = r["Name"]
    if "pf_gemm" in n and "lds" in n or "pf_gemm" in n: print("  %-48s %8.2f us" % (n.replace("l2k::", "").replace("(PfArgs)", "")[:48], float(r["AverageNs"]) / 1e3))
PY
}
run base L2_PF_LDS=0
run nw8 L2_PF_LDS=0 L2_PF_NW_QKV=8 L2_PF_NW_WO=8 L2_PF_NW_W13=8 L2_PF_NW_W2=8
run lds L2_PF_LDS=1
run nolds L2_PF_LDS=0
