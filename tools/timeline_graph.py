"""Timeline of one token's GEMV launches UNDER GRAPH REPLAY from the diagnostic (L2_STAMPS) build: every workgroup of a phase launch
records {start, end} on the 100 MHz clock all XCDs share; per launch (enqueue order): first start, last end, and the time since the
previous phase launch's last end (= boundaries + whatever launch without stamps -- attention -- ran in between).

  python tools/timeline_graph.py [config] [position]     (the run must stay within one step level: one captured graph)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools")); import diag_lib; diag_lib.use()    # builds gpurun_out/diag/libllama2hip_stamps.so on demand
os.environ["L2_USE_GRAPH"] = "1"
os.environ.setdefault("L2_TEST_HOOKS", "1")
import numpy as np
from llama2_ts_amd import configs, runtime
name = sys.argv[1] if len(sys.argv) > 1 else "stories110M"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 100
hdr = configs.header(name)
ctx = runtime.Context(hdr); ctx.synth_fill(1)
L = runtime.lib(); L.l2_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
acc = {}
REP = 8
for rep in range(REP):
    ctx.decode_greedy(1, 0, F + rep)
    buf = np.zeros(66 * 108 + 64 * 2048, dtype=np.uint64)
    assert L.l2_debug_stamps(ctx._h, buf.ctypes.data, buf.size) == 0
    wg = buf[66 * 108:].reshape(64, 1024, 2).astype(np.int64)
    rows = []
    for slot in range(64):
        t = wg[slot]; t = t[t[:, 0] > 0]
        if not len(t): continue
        rows.append((slot, t[:, 0].min(), (t[:, 0] + (t[:, 1] & 0xffffffff)).max(), len(t), np.median(t[:, 1] & 0xffffffff)))
    # the last token's launches: those whose start lies within 2 ms of the latest start
    latest = max(r[1] for r in rows)
    rows = sorted([r for r in rows if latest - r[1] < 200000], key=lambda r: r[1])
    prev_end = None
    for i, (slot, st, en, n, med) in enumerate(rows):
        acc.setdefault(i, []).append((slot, n, (en - st) / 100.0, med / 100.0, (st - prev_end) / 100.0 if prev_end else 0.0))
        prev_end = en
print("config %s, token at position %d, graph replay, mean of %d runs (us)" % (name, F - 1, REP))
print("%4s %5s %5s %10s %10s %12s" % ("#", "slot", "wgs", "span", "median wg", "since prev"))
tot = 0.0
for i in sorted(acc):
    a = np.array([x[2:] for x in acc[i]])
    print("%4d %5d %5d %10.2f %10.2f %12.2f" % (i, acc[i][0][0], acc[i][0][1], a[:, 0].mean(), a[:, 1].mean(), a[:, 2].mean()))
# in-kernel stamps (cycles since the wave's stamp 0) of the launches of the first layer: waves 0, 1, last of the first / middle / last workgroup
st = buf[:64 * 108].reshape(64, 3, 3, 12).astype(np.int64)
print("stamps of the last run, slots 0..4 (cycles since stamp 0 of the wave; ids as in tools/stamps.py)")
for slot in range(5):
    for w, wn in enumerate(("first", "mid", "last")):
        for wv in range(3):
            t = st[slot, w, wv]
            if t[0] == 0: continue
            print("slot %d %-5s %-2s " % (slot, wn, ("w0", "w1", "wL")[wv]) + " ".join("%6s" % (str(int(t[k] - t[0])) if t[k] else "-") for k in range(1, 10)))
