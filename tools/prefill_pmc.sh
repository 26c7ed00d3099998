# Where do the cycles of the prompt-ingestion GEMMs go?  SQ counters of the shipped and the register-blocked kernels (7B width, 64 tokens),
# separate --pmc passes with kernel trace only (MI355X guide).  bash tools/prefill_pmc.sh -> gpurun_out/pfpmc/summary.txt
export L2_TEST_HOOKS=1   # the development switches below only exist behind this gate
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pfpmc
rm -rf $out; mkdir -p $out
rocprofv3 -L > $out/counters.txt 2>&1
i=0
for ctr in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "FETCH_SIZE" ; do
  i=$((i+1))
  for v in base pf3; do
    if [ $v = base ]; then export L2_PF3=0; else export L2_PF3=1 L2_PF3_RT_QKV=3 L2_PF3_NW_QKV=4 L2_PF3_NW_WO=4 L2_PF3_RT_W13=1 L2_PF3_NW_W13=4 L2_PF3_NW_W2=4; fi
    PF_TOKENS=64 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/p${i}_$v -o p -- python3 tools/pf_target.py > $out/p${i}_$v.log 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections
out = "gpurun_out/pfpmc"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "pf_gemm" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0].replace("void l2k::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as f:
    for k in sorted(acc):
        f.write(k + "\n")
        for c in sorted(acc[k]):
            v = acc[k][c]; v = v[2:] if len(v) > 4 else v
            f.write("   %-36s %16.1f\n" % (c, sum(v) / len(v)))
print(open(out + "/summary.txt").read())
PY
