"""Latency anatomy of the GEMV phase kernels from the diagnostic (L2_STAMPS) build: shader-clock stamps of wave 0
of the first / middle / last workgroup of every launch of one token, printed as cycles since the wave's first stamp.

  python tools/stamps.py [config] [position]

Stamp ids (kernels.hip.h): streaming form 0 start, 1 x + first weights requested, 2 x in LDS, 3 sum(x^2) reduced,
4 normalised + barrier, 5 first batch consumed, 6 rows reduced, 7 epilogue; latency form 0 start, 1 everything
requested, 2 sum(x^2) reduced (wave 0), 3 barrier passed, 5 dot products done, 6 rows reduced, 7 epilogue."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools")); import diag_lib; diag_lib.use()    # builds gpurun_out/diag/libllama2hip_stamps.so on demand
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
per_tok = 4 * cfg.n_layers + 1            # phase-kernel launches per token (attention stamps: stamps_attn.py)
buf = np.zeros(66 * 108 + 64 * 2048, dtype=np.uint64)
L = runtime.lib()
L.l2_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
assert L.l2_debug_stamps(ctx._h, buf.ctypes.data, buf.size) == 0
wg = buf[66 * 108:].reshape(64, 1024, 2).astype(np.int64)
buf = buf[:64 * 108].reshape(64, 3, 3, 12).astype(np.int64)
names = ["qkv", "wo", "w13", "w2"]
first = per_tok * (F - 1)
print("config", name, "pos", F - 1, "(cycles since the wave's stamp 0; 1000 cycles ~ 0.42-0.48 us; w0 = wave 0 (the x wave of the latency form), w1 = wave 1)")
print("%-10s %-8s " % ("kernel", "wg/wave") + " ".join("%7s" % ("s%d" % k) for k in range(1, 12)))
if os.environ.get("STAMPS_WG"):   # when every workgroup of a launch started and ended (100 MHz clock shared by the XCDs; streaming form only)
    print("per launch: workgroups, then start and end times in us after the first start: min / median / max")
    for j in list(range(0, 8)) + [per_tok - 1]:
        slot = (first + j) % 64
        nm = "cls" if j == per_tok - 1 else "%s.l%d" % (names[j % 4], j // 4)
        t = wg[slot]; t = t[t[:, 0] > 0]
        if not len(t):
            continue
        t = t[np.abs(t[:, 0] - np.median(t[:, 0])) < 100000]      # a smaller grid leaves an older launch's entries in the slot
        t0 = t[:, 0].min(); st = (t[:, 0] - t0) / 100.0; life = (t[:, 1] & 0xffffffff) / 100.0; en = st + life
        xcc = (t[:, 1] >> 32) & 0xf; hw = (t[:, 1] >> 44) & 0xffff      # HW_ID bits 8..: cu_id (4), sh_id (1), se_id (3)
        print("%-8s %4d wgs  start %5.2f / %5.2f / %5.2f   end %6.2f / %6.2f / %6.2f   lifetime %6.2f / %6.2f / %6.2f" % (
            nm, len(t), st.min(), np.median(st), st.max(), en.min(), np.median(en), en.max(), life.min(), np.median(life), life.max()))
        if os.environ.get("STAMPS_WG") == "3" and nm in ("w13.l0", "qkv.l0"):
            full = wg[slot]
            for b in list(range(0, 12)) + list(range(250, 262)) + list(range(452, 459)):
                if full[b, 0]: print("          block %3d -> xcc %d hw_id 0x%04x lifetime %.1f" % (b, (full[b, 1] >> 32) & 0xf, (full[b, 1] >> 44) & 0xffff, (full[b, 1] & 0xffffffff) / 100.0))
        if os.environ.get("STAMPS_WG") == "2":
            print("          median lifetime by XCD: " + " ".join("%d:%.1f(%d)" % (x, np.median(life[xcc == x]), (xcc == x).sum()) for x in sorted(set(xcc.tolist()))))
            cu = hw & 0xff
            per_cu = {}
            for k in range(len(t)):
                per_cu.setdefault((int(xcc[k]), int(cu[k])), []).append(life[k])
            ones = [v[0] for v in per_cu.values() if len(v) == 1]; twos = [x for v in per_cu.values() if len(v) == 2 for x in v]; more = [x for v in per_cu.values() if len(v) > 2 for x in v]
            print("          CUs holding 1 / 2 / more workgroups: %d / %d / %d; median lifetime %.1f / %.1f / %.1f" % (
                len(ones), len(twos) // 2, len(per_cu) - len(ones) - len(twos) // 2, np.median(ones) if ones else 0, np.median(twos) if twos else 0, np.median(more) if more else 0))
    if os.environ.get("STAMPS_WG") == "4":   # is a workgroup that is slow in one launch slow in the next launch of the same kind?
        for a_, b_ in ((0, 4), (1, 5), (2, 6), (3, 7)):
            la = (wg[(first + a_) % 64][:, 1] & 0xffffffff) / 100.0; lb = (wg[(first + b_) % 64][:, 1] & 0xffffffff) / 100.0
            ok = (wg[(first + a_) % 64][:, 0] > 0) & (wg[(first + b_) % 64][:, 0] > 0) & (la > 0.5 * np.median(la[la > 0])) & (lb > 0.5 * np.median(lb[lb > 0])) & (la < 2 * np.median(la[la > 0])) & (lb < 2 * np.median(lb[lb > 0]))
            nb = 459 if a_ == 2 else 512
            ok[nb:] = False
            print("%s layer 0 vs layer 1: correlation of the workgroups' lifetimes %.2f (n = %d, std %.2f / %.2f us)" % (names[a_], np.corrcoef(la[ok], lb[ok])[0, 1], ok.sum(), la[ok].std(), lb[ok].std()))
    sys.exit(0)
if os.environ.get("STAMPS_ABS"):   # start / end of the three stamped workgroups on one clock (s_memtime), relative to the earliest start of the launch
    print("absolute: start and end (stamp 7) of wave 0 of the first / middle / last workgroup, cycles after the earliest of the three starts")
    for j in list(range(0, 8)) + [per_tok - 1]:
        slot = (first + j) % 64
        nm = "cls" if j == per_tok - 1 else "%s.l%d" % (names[j % 4], j // 4)
        t0 = min(int(buf[slot, w, 0, 0]) for w in range(3) if buf[slot, w, 0, 0])
        print("%-8s " % nm + "  ".join("%s start %6d end %7d" % (wn, int(buf[slot, w, 0, 0]) - t0, int(buf[slot, w, 0, 7]) - t0) for w, wn in enumerate(("first", "mid", "last"))))
    sys.exit(0)
for j in list(range(0, 8)) + [per_tok - 1]:
    slot = (first + j) % 64
    nm = "cls" if j == per_tok - 1 else "%s.l%d" % (names[j % 4], j // 4)
    for w, wn in enumerate(("first", "mid", "last")):
        for wv in range(3):
            t = buf[slot, w, wv]
            if t[0] == 0:
                continue
            print("%-10s %-8s " % (nm, "%s/%s" % (wn, ("w0", "w1", "wL")[wv])) + " ".join("%7s" % (str(int(t[k] - t[0])) if t[k] else "-") for k in range(1, 12)))
