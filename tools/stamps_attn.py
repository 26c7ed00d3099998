"""Latency anatomy of attn_kernel (diagnostic L2_STAMPS build): cycles between stages, wave 0 of head 0 / mid / last."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["L2_LIB_PATH"] = os.path.join(ROOT, "llama2.ts_amd", "lib", "libllama2hip_stamps.so")
os.environ["L2_USE_GRAPH"] = "0"
import numpy as np
from llama2_ts_amd import configs, runtime
name = sys.argv[1]; F = int(sys.argv[2])
ctx = runtime.Context(configs.header(name)); ctx.synth_fill(1)
tok = 1
for pos in range(F):
    tok = int(np.argmax(ctx.forward(tok, pos)))
buf = np.zeros(66 * 36, dtype=np.uint64)
L = runtime.lib(); L.l2_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
assert L.l2_debug_stamps(ctx._h, buf.ctypes.data, buf.size) == 0
t = buf[64 * 36:65 * 36].reshape(3, 12).astype(np.int64)
labels = ["stage q/k/v", "scores", "softmax", "values", "merge+store"]
print(name, "pos", F - 1, "last layer (cycles)")
for w, wn in enumerate(("head0", "mid", "last")):
    print("%-6s" % wn, " ".join("%s %6d" % (l, t[w][k + 1] - t[w][k]) for k, l in enumerate(labels)), " total", t[w][5] - t[w][0])
w = buf[65 * 36:66 * 36].reshape(3, 12).astype(np.int64)
if w[0][0]:
    a0 = t[0][0]
    print("fused attention + wo launch, cycles relative to head 0's start:")
    for k, wn in enumerate(("head0", "mid", "last")):
        print("  attn %-5s start %6d  xb stored %6d" % (wn, t[k][0] - a0, t[k][5] - a0))
    wl = ["start", "weights requested", "weights landed", "flag seen", "xb in LDS", "rows done"]
    for k, wn in enumerate(("first", "mid", "last")):
        print("  wo   %-5s " % wn + "  ".join("%s %6d" % (l, w[k][j] - a0) for j, l in enumerate(wl)))
