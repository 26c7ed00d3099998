"""One rank's shard of the tensor-parallel step alone on this GPU (l2_create_tp with L2_TP_SOLO_ID: exchange kernels against the rank's
own inbox): decode a few tokens.  Meant to run under rocprofv3 --kernel-trace --stats with L2_USE_GRAPH=0 L2_PROFILE_SYNC=1:
  python tools/tp_solo_step.py <G> [config] [tokens]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llama2_ts_amd import configs, runtime
G = int(sys.argv[1]); name = sys.argv[2] if len(sys.argv) > 2 else "llama2_7b"; n = int(sys.argv[3]) if len(sys.argv) > 3 else 24
hdr = configs.header(name)
c = runtime.Context(hdr, tp_rank=0, tp_size=G, nccl_id=runtime.TP_SOLO_ID) if G > 1 else runtime.Context(hdr)
c.synth_fill(1)
c.bench_decode(1, 0, n)
ms = c.bench_decode(1, 0, n)
print("G=%d %s: %.4f ms per token (%s)" % (G, name, ms / n, c.tp_mode()))
c.close()
