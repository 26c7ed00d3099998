#!/bin/bash
# First contact with a multi-GPU node (no such node has ever run this code): every step a FRESH process, in the order that tells the
# most the soonest -- the two-GPU tests (RCCL with two ranks, the peer mapping across GPUs), the 2-layer 7B-width model on two ranks
# (a golden of the real reference exists for it), then the full Llama-2-7B at 2, 4 and 8 ranks.  Each bench.py run proves every way of
# forming the group before it times anything and says in its `note` which formation steps failed and why; the digest at the end puts
# l2_tp_mode, those notes, the parity of the timed run and measured-against-predicted side by side.
#   bash tools/first_contact.sh [transcript]         (default: gpurun_out/first_contact.txt)
OUT=${1:-gpurun_out/first_contact.txt}
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
N=$(python -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 1)
step() { echo "== $1" | tee -a "$OUT"; shift; timeout 1800 "$@" 2>&1 | tail -40 | tee -a "$OUT"; }
step "pytest: two-GPU group over RCCL and xGMI, a context on another device than the thread's" python -m pytest tests/test_tp_gpu.py -q -m gpu -k "two_gpu or other_than_the_threads"
[ "$N" -ge 2 ] && step "bench --gpus 2 --config llama2_7b_L2 (golden: 2048 steps of the real reference)" python bench.py --gpus 2 --config llama2_7b_L2 --steps 64 --warmup 8
for G in 2 4 8; do
  [ "$N" -ge "$G" ] && step "bench --gpus $G (full Llama-2-7B)" python bench.py --gpus $G --steps 64 --warmup 8
done
[ "$N" -ge 2 ] && L2_TP_FENCED=1 step "bench --gpus 2 with L2_TP_FENCED=1 (the flag exchange with system-scope fences: the form to fall back to if ranks diverge)" python bench.py --gpus 2 --steps 64 --warmup 8
echo; echo "---- digest"; python tools/first_contact_summary.py "$OUT"
