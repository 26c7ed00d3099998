"""Timeline of the fused QKV + attention launch (diagnostic L2_STAMPS build): the last layer's launch of one token, every stamped
wave on ONE time axis (each wave's start on the 100 MHz clock the XCDs share + its shader-clock stamps at 2.4 GHz), microseconds
after the earliest start.

  python tools/stamps_fused.py [config] [position]

QKV role (first / middle workgroup; w0 = the x wave, w1 / w7 compute waves): requested, barrier passed, dots done, rows reduced,
epilogue stores (cache rows + hand-off granules) issued.  Attention role (the last workgroup = last head; w0, w1, w7 = the wave
that scores row pos): tiles requested, granules arrived, scores, softmax, values, stored."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools")); import diag_lib; diag_lib.use()    # builds gpurun_out/diag/libllama2hip_stamps.so on demand
os.environ["L2_USE_GRAPH"] = "0"
os.environ.setdefault("L2_TEST_HOOKS", "1")
import numpy as np
from llama2_ts_amd import configs, runtime
name = sys.argv[1] if len(sys.argv) > 1 else "stories110M"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 100
hdr = configs.header(name); cfg = runtime.Config(hdr)
ctx = runtime.Context(hdr); ctx.synth_fill(1)
toks = ctx.decode_greedy(1, 0, F)          # eager launches (L2_USE_GRAPH=0): the stamp slots rotate per launch
per_tok = 4 * cfg.n_layers + 1
buf = np.zeros(66 * 108 + 64 * 2048, dtype=np.uint64)
L = runtime.lib()
L.l2_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
assert L.l2_debug_stamps(ctx._h, buf.ctypes.data, buf.size) == 0
ph = buf[:64 * 108].reshape(64, 3, 3, 12).astype(np.int64)
at = buf[64 * 108:65 * 108].reshape(3, 3, 12).astype(np.int64)
slot = (per_tok * (F - 1) + 4 * (cfg.n_layers - 1)) % 64          # the last token's last layer: its fused launch
GHZ = 2.4
rows = []
for w, wn in ((0, "qkv first"), (1, "qkv mid")):
    for wv, wvn in ((0, "w0 (x)"), (1, "w1"), (2, "w7")):
        t = ph[slot, w, wv]
        if t[0]: rows.append((wn + " " + wvn, t, {1: "requested", 3: "barrier", 5: "dots", 6: "reduced", 7: "stores issued"}))
for wv, wvn in ((0, "w0"), (1, "w1"), (2, "w7 (row pos)")):
    t = at[2, wv]
    if t[0]: rows.append(("attn last " + wvn, t, {1: "requested", 6: "granules", 2: "scores", 3: "softmax", 4: "values", 5: "stored"}))
if not rows:
    print("no stamps (is the fused launch in use for this shape?)"); sys.exit(1)
base = min(int(t[11]) for _, t, _ in rows)
print(name, "pos", F - 1, "last layer; microseconds after the earliest stamped wave's start")
for nm, t, lab in rows:
    st = (int(t[11]) - base) / 100.0
    print("%-24s start %5.2f  " % (nm, st) + "  ".join("%s %5.2f" % (l, st + (int(t[k]) - int(t[0])) / (GHZ * 1e3)) for k, l in sorted(lab.items(), key=lambda kv: int(t[kv[0]])) if t[k]))
ctx.close()
