"""Target for the rocprofv3 --pmc passes: a few launches of each weight-streaming phase kernel on 7B-width
matrices (2-layer model), nothing else on the GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llama2_ts_amd import configs, runtime
hdr = configs.header(sys.argv[1] if len(sys.argv) > 1 else "llama2_7b_L2")
ctx = runtime.Context(hdr); ctx.synth_fill(1)
ctx.forward(1, 0)
for kind in (runtime.T_WQ, runtime.T_WO, runtime.T_W1, runtime.T_W2, runtime.T_WCLS):
    ctx.bench_gemv(kind, 1, 4)
ctx.close()
