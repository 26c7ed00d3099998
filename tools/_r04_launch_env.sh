#!/bin/bash
# the replayed-graph floor under runtime knobs (strings of libamdhip64.so): does any of them move the 1.65 us per kernel node?
cd "$(dirname "$0")" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mbl microbench_launch.hip || exit 1
run() { echo "== $*"; env "$@" /tmp/mbl | grep -E "grid +256 block 256|grid +1 block  64"; }
run X=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=256
run AMD_OPT_FLUSH=0
run ROC_USE_FGS_KERNARG=0
run ROC_USE_FGS_KERNARG=1
run DEBUG_HIP_KERNARG_COPY_OPT=0
run DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0
run DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1
run HIP_FORCE_DEV_KERNARG=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run GPU_FLUSH_ON_EXECUTION=1
run AMD_DIRECT_DISPATCH=0
run ROC_ACTIVE_WAIT_TIMEOUT=0
run HSA_ENABLE_INTERRUPT=0
