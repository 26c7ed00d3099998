"""Which workgroups of a streaming launch finish late?  Per launch of one token (L2_STAMPS build, eager launches): lifetime of the
workgroups by XCD, by shader engine and by position in the grid.

  python tools/stamps_xcd.py [config] [position]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools")); import diag_lib; diag_lib.use()    # builds gpurun_out/diag/libllama2hip_stamps.so on demand
os.environ["L2_USE_GRAPH"] = "0"
os.environ.setdefault("L2_TEST_HOOKS", "1")
import numpy as np
from llama2_ts_amd import configs, runtime
name = sys.argv[1] if len(sys.argv) > 1 else "llama2_7b_L2"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 20
hdr = configs.header(name); cfg = runtime.Config(hdr)
ctx = runtime.Context(hdr); ctx.synth_fill(1)
tok = 1
for pos in range(F):
    tok = int(np.argmax(ctx.forward(tok, pos)))
if os.environ.get("XCD_ONLY"):      # compact: median end per XCD for every launch of several consecutive tokens
    L = runtime.lib(); L.l2_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    per_tok = 4 * cfg.n_layers + 1; names = ["qkv", "wo", "w13", "w2"]
    for pos in range(F, F + int(os.environ["XCD_ONLY"])):
        tok = int(np.argmax(ctx.forward(tok, pos)))
        buf = np.zeros(66 * 108 + 64 * 2048, dtype=np.uint64)
        assert L.l2_debug_stamps(ctx._h, buf.ctypes.data, buf.size) == 0
        wg = buf[66 * 108:].reshape(64, 1024, 2).astype(np.int64)
        for j in range(per_tok):
            slot = (per_tok * pos + j) % 64
            nm = "cls" if j == per_tok - 1 else "%s.l%d" % (names[j % 4], j // 4)
            t = wg[slot]; t = t[t[:, 0] > 0]
            t = t[np.abs(t[:, 0] - np.median(t[:, 0])) < 100000]
            en = ((t[:, 0] - t[:, 0].min()) + (t[:, 1] & 0xffffffff)) / 100.0; xcc = (t[:, 1] >> 32) & 0xf
            print("pos %3d %-7s max %5.1f  by XCD: " % (pos, nm, en.max()) + " ".join("%5.1f" % np.median(en[xcc == x]) for x in range(8)))
    sys.exit(0)
per_tok = 4 * cfg.n_layers + 1
buf = np.zeros(66 * 108 + 64 * 2048, dtype=np.uint64)
L = runtime.lib()
L.l2_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
assert L.l2_debug_stamps(ctx._h, buf.ctypes.data, buf.size) == 0
wg = buf[66 * 108:].reshape(64, 1024, 2).astype(np.int64)
names = ["qkv", "wo", "w13", "w2"]
first = per_tok * (F - 1)
for j in list(range(4, 8)) + [per_tok - 1]:
    slot = (first + j) % 64
    nm = "cls" if j == per_tok - 1 else "%s.l%d" % (names[j % 4], j // 4)
    full = wg[slot]
    idx = np.nonzero(full[:, 0] > 0)[0]
    t = full[idx]
    ok = np.abs(t[:, 0] - np.median(t[:, 0])) < 100000
    idx, t = idx[ok], t[ok]
    t0 = t[:, 0].min(); st = (t[:, 0] - t0) / 100.0; life = (t[:, 1] & 0xffffffff) / 100.0; en = st + life
    xcc = (t[:, 1] >> 32) & 0xf; hw = (t[:, 1] >> 44) & 0xffff
    cu = hw & 0xf; sh = (hw >> 4) & 1; se = (hw >> 5) & 7
    print("%-7s %d workgroups, end min / median / max %.1f / %.1f / %.1f us" % (nm, len(t), en.min(), np.median(en), en.max()))
    print("   by XCD   : " + "  ".join("%d: %.1f (%d)" % (x, np.median(en[xcc == x]), (xcc == x).sum()) for x in sorted(set(xcc))))
    print("   by SE    : " + "  ".join("%d: %.1f" % (x, np.median(en[se == x])) for x in sorted(set(se))))
    print("   by CU id : " + "  ".join("%d: %.1f" % (x, np.median(en[cu == x])) for x in sorted(set(cu))))
    q = len(idx) // 8
    print("   by block index (eighths of the grid): " + "  ".join("%.1f" % np.median(en[(idx >= k * q) & (idx < (k + 1) * q)]) for k in range(8)))
    per_cu = {}
    for i in range(len(t)):
        per_cu.setdefault((int(xcc[i]), int(hw[i])), []).append(en[i])
    ones = [v[0] for v in per_cu.values() if len(v) == 1]; twos = [max(v) for v in per_cu.values() if len(v) == 2]
    print("   CUs with one workgroup: %d (median end %.1f)   with two: %d (median end of the later one %.1f)   CUs used: %d" % (
        len(ones), np.median(ones) if ones else 0, len(twos), np.median(twos) if twos else 0, len(per_cu)))
    late = np.argsort(en)[-8:]
    print("   the 8 latest: " + "  ".join("b%d x%d se%d cu%d %.1f" % (idx[i], xcc[i], se[i], cu[i], en[i]) for i in late))
