"""Per-phase timeline of one chain launch (diagnostic L2_STAMPS build): first/last workgroup start and end,
in microseconds from the launch's first workgroup (s_memrealtime, 100 MHz)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["L2_LIB_PATH"] = os.path.join(ROOT, "llama2.ts_amd", "lib", "libllama2hip_stamps.so")
os.environ["L2_USE_GRAPH"] = "0"; os.environ["L2_CHAIN"] = "1"
import numpy as np
from llama2_ts_amd import configs, runtime
name = sys.argv[1]; F = int(sys.argv[2]); nshow = int(sys.argv[3]) if len(sys.argv) > 3 else 12
ctx = runtime.Context(configs.header(name)); ctx.synth_fill(1)
tok = 1
for pos in range(F):
    tok = int(np.argmax(ctx.forward(tok, pos)))
L = runtime.lib()
nb = C.c_int(); off = (C.c_int * 6)(); nl = C.c_int(); cls = (C.c_int * 2)()
L.l2_debug_chain_timeline(None, C.byref(nb), off, C.byref(nl), cls)
buf = np.zeros(nb.value * 2, dtype=np.uint64)
L.l2_debug_chain_timeline.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
L.l2_debug_chain_timeline(buf.ctypes.data, C.byref(nb), off, C.byref(nl), cls)
t = buf.reshape(-1, 2).astype(np.int64)
t0 = t[:, 0].min()
us = (t - t0) / 100.0
names = ["qkv", "attn", "wo", "w13", "w2"]
per = off[5]
print(name, "pos", F - 1, "blocks", nb.value, "per layer", per, list(off))
print("%-8s %6s | %8s %8s | %8s %8s | %s" % ("phase", "wgs", "start0", "startN", "end0", "endN", "span_us"))
rows = []
for l in range(nl.value):
    for r in range(5):
        a, b = l * per + off[r], l * per + off[r + 1]
        rows.append(("%s.%d" % (names[r], l), us[a:b]))
rows.append(("cls", us[cls[0]:cls[0] + cls[1]]))
prev_end = 0.0
for nm, u in rows[:nshow] + rows[-3:]:
    print("%-8s %6d | %8.2f %8.2f | %8.2f %8.2f | %6.2f   gap-from-prev-end %.2f" % (nm, len(u), u[:, 0].min(), u[:, 0].max(), u[:, 1].min(), u[:, 1].max(), u[:, 1].max() - u[:, 0].min(), u[:, 1].min() - prev_end))
    prev_end = u[:, 1].max()
print("total", us[:, 1].max())
