# per-kernel times of the prompt-ingestion GEMMs, 7B width: the 16-row-tile kernels (L2_PF3=0) against the register-blocked ones with 1, 2 and 4
# 64-token chunks per launch (PF_TOKENS = 64 / 128 / 256).  bash tools/prefill_variants.sh   (GPU box)
export L2_TEST_HOOKS=1   # the development switches below only exist behind this gate
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {
  tag=$1; shift
  rm -rf gpurun_out/pfv_$tag
  ( export "$@"; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pfv_$tag -o p -- python3 tools/pf_target.py > /dev/null 2>&1 )
  echo "== $tag ($*)"
  python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/pfv_$tag/p_kernel_stats.csv")):
    n = r["Name"]
    if "pf_" in n: print("  %-52s %8.2f us x %s" % (n.replace("l2k::", "").replace("(l2k::PfArgs)", "").replace("void ", "")[:52], float(r["AverageNs"]) / 1e3, r["Calls"]))
PY
}
run old64 L2_PF3=0 PF_TOKENS=64
run new64 L2_PF3=1 PF_TOKENS=64
run new128 L2_PF3=1 PF_TOKENS=128
run new256 L2_PF3=1 PF_TOKENS=256
