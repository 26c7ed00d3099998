"""Latency anatomy of the attention tile kernel (diagnostic L2_STAMPS build): cycles since the wave's first stamp,
wave 0 of the first / middle / last workgroup of the last layer's launch.
Stamps (attention.hip.h): 1 K / q / V requested, 2 scores in LDS, 3 probabilities in LDS, 4 value partials in LDS, 5 xb stored."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools")); import diag_lib; diag_lib.use()    # builds gpurun_out/diag/libllama2hip_stamps.so on demand
os.environ["L2_USE_GRAPH"] = "0"
import numpy as np
from llama2_ts_amd import configs, runtime
name = sys.argv[1]; F = int(sys.argv[2])
ctx = runtime.Context(configs.header(name)); ctx.synth_fill(1)
tok = 1
for pos in range(F):
    tok = int(np.argmax(ctx.forward(tok, pos)))
buf = np.zeros(66 * 108, dtype=np.uint64)
L = runtime.lib(); L.l2_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
assert L.l2_debug_stamps(ctx._h, buf.ctypes.data, buf.size) == 0
t = buf[64 * 108:65 * 108].reshape(3, 3, 12).astype(np.int64)[:, 0]
labels = ["requested", "scores", "softmax", "values", "stored"]
print(name, "pos", F - 1, "last layer (cycles since stamp 0)")
for w, wn in enumerate(("first", "mid", "last")):
    if t[w][0]:
        print("%-6s" % wn, " ".join("%s %6s" % (l, str(int(t[w][k + 1] - t[w][0])) if t[w][k + 1] else "-") for k, l in enumerate(labels)))
