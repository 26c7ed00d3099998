"""In situ against back to back: the same streaming launch (w1/w3 of layer 1) inside a decode step and launched alone five times in a
row (l2_bench_gemv), L2_STAMPS build: wave 0's prologue stamps (cycles) and when the workgroups end (us).

  python tools/stamps_isolated.py [config]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools")); import diag_lib; diag_lib.use()    # builds gpurun_out/diag/libllama2hip_stamps.so on demand
os.environ["L2_USE_GRAPH"] = "0"
os.environ.setdefault("L2_TEST_HOOKS", "1")
import numpy as np
from llama2_ts_amd import configs, runtime
name = sys.argv[1] if len(sys.argv) > 1 else "llama2_7b_L2"
hdr = configs.header(name); cfg = runtime.Config(hdr)
ctx = runtime.Context(hdr); ctx.synth_fill(1)
L = runtime.lib(); L.l2_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
per_tok = 4 * cfg.n_layers + 1
launches = 0

def read(slot):
    buf = np.zeros(66 * 108 + 64 * 2048, dtype=np.uint64)
    assert L.l2_debug_stamps(ctx._h, buf.ctypes.data, buf.size) == 0
    wg = buf[66 * 108:].reshape(64, 1024, 2).astype(np.int64)[slot % 64]
    st = buf[:64 * 108].reshape(64, 3, 3, 12).astype(np.int64)[slot % 64]
    t = wg[wg[:, 0] > 0]
    t = t[np.abs(t[:, 0] - np.median(t[:, 0])) < 100000]
    en = ((t[:, 0] - t[:, 0].min()) + (t[:, 1] & 0xffffffff)) / 100.0
    return st, en

def show(tag, slot):
    st, en = read(slot)
    print("%-34s workgroups end %.1f / %.1f / %.1f us (min / median / max)" % (tag, en.min(), np.median(en), en.max()))
    for w, wn in enumerate(["first", "mid", "last"]):
        s = st[w, 0]
        print("      %-5s wave 0: requested %5d  x in LDS %5d  sum %5d  normalised + barrier %5d  last batch %6d  epilogue %6d" % (wn, s[1] - s[0], s[2] - s[0], s[3] - s[0], s[4] - s[0], s[5] - s[0], s[7] - s[0]))

tok = 1
for pos in range(12):
    tok = int(np.argmax(ctx.forward(tok, pos))); launches += per_tok
for which, kind, j in (("w1/w3", runtime.T_W1, 2), ("wq/wk/wv", runtime.T_WQ, 0), ("w2", runtime.T_W2, 3), ("wo", runtime.T_WO, 1)):
    tok = int(np.argmax(ctx.forward(tok, 12))); launches += per_tok
    show("%s of layer 1, inside a decode step" % which, launches - per_tok + 4 + j)
    ctx.bench_gemv(kind, 1, 5); launches += 7          # 2 warm-up + 5
    show("%s of layer 1, alone, 7th in a row" % which, launches - 1)
