"""Latency anatomy of the phase kernels from the diagnostic (L2_STAMPS) build: shader-clock deltas between
stages for wave 0 of the first / middle / last workgroup of every launch of one token."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["L2_LIB_PATH"] = os.path.join(ROOT, "llama2.ts_amd", "lib", "libllama2hip_stamps.so")
os.environ["L2_USE_GRAPH"] = "0"
import numpy as np
from llama2_ts_amd import configs, runtime
name = sys.argv[1] if len(sys.argv) > 1 else "stories110M"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 20
hdr = configs.header(name); cfg = runtime.Config(hdr)
ctx = runtime.Context(hdr); ctx.synth_fill(1)
tok = 1
for pos in range(F):
    tok = int(np.argmax(ctx.forward(tok, pos)))
per_tok = 4 * cfg.n_layers + 1            # phase-kernel launches per token (attention has no stamps)
buf = np.zeros(64 * 36, dtype=np.uint64)
L = runtime.lib()
L.l2_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
assert L.l2_debug_stamps(ctx._h, buf.ctypes.data, buf.size) == 0
buf = buf.reshape(64, 3, 12).astype(np.int64)
names = ["qkv", "wo", "w13", "w2"]
labels = ["x+w issued", "x landed", "x in LDS", "ss reduced", "norm+barrier", "weights consumed", "rows reduced", "epilogue"]
first = per_tok * (F - 1)
print("config", name, "pos", F - 1, "(cycles; ~2.1-2.4 GHz => 1000 cycles ~ 0.45 us)")
print("%-10s %-5s " % ("kernel", "wg") + " ".join("%16s" % l for l in labels) + "   total")
for j in list(range(0, 8)) + [per_tok - 1]:
    slot = (first + j) % 64
    nm = "cls" if j == per_tok - 1 else "%s.l%d" % (names[j % 4], j // 4)
    for w, wn in enumerate(("first", "mid", "last")):
        t = buf[slot, w]
        if t[0] == 0:
            continue
        d = [int(t[k + 1] - t[k]) if t[k + 1] and t[k] else 0 for k in range(7)]
        has_norm = t[3] != 0
        if not has_norm:   # copy modes have no stamp 3
            d[2] = 0; d[3] = int(t[4] - t[2])
        print("%-10s %-5s " % (nm, wn) + " ".join("%16d" % v for v in [0] + d)[17:] + "   %6d" % int(t[7] - t[0]))
