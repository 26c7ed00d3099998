import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from llama2_ts_amd import configs, runtime
name = sys.argv[1]
hdr = configs.header(name); cfg = runtime.Config(hdr)
ctx = runtime.Context(hdr); ctx.synth_fill(1)
rng = np.random.default_rng(0)
ctx.forward(1, 0)
t0 = time.perf_counter()
for p in range(16): ctx.forward(1, p)
tf = (time.perf_counter() - t0) / 16
print(name, "one l2_forward: %.2f ms" % (tf * 1e3))
for n in (1, 2, 4, 8, 16, 24, 32, 48, 64, 96, 128):
    toks = rng.integers(2, cfg.vocab_size, n).astype(np.int32)
    ctx.prefill(toks, 0)
    t0 = time.perf_counter(); ctx.prefill(toks, 0); ctx.prefill(toks, 0); t1 = time.perf_counter()
    ms = (t1 - t0) / 2 * 1e3
    print("  n=%3d  prefill %.2f ms  (%.2f ms per token, token-by-token %.2f ms, %.1fx)" % (n, ms, ms / n, tf * n * 1e3, tf * n * 1e3 / ms))
