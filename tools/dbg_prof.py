import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llama2_ts_amd import configs, runtime
import numpy as np
mode, cfgname = sys.argv[1], sys.argv[2]
hdr = configs.header(cfgname)
ctx = runtime.Context(hdr); ctx.synth_fill(1)
print("created", flush=True)
if "g" in mode:
    print(ctx.bench_decode(1, 0, int(sys.argv[3])), flush=True)
if "f" in mode:
    print(np.argmax(ctx.forward(1, 0)), flush=True)
    print(np.argmax(ctx.forward(1, 1)), flush=True)
if "k" in mode:
    print(ctx.bench_gemv(runtime.T_W1, 0, 10), flush=True)
print("done", flush=True)
