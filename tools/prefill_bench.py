"""Prompt ingestion: l2_prefill (fp64-MFMA chunks of up to 64 tokens) vs one l2_forward per prompt token."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from llama2_ts_amd import configs, runtime
name = sys.argv[1]
hdr = configs.header(name); cfg = runtime.Config(hdr)
ctx = runtime.Context(hdr); ctx.synth_fill(1)
rng = np.random.default_rng(0)
for n in (16, 64, 256):
    n = min(n, cfg.seq_len)
    toks = rng.integers(2, cfg.vocab_size, n).astype(np.int32)
    ctx.prefill(toks[:min(n, 64)], 0)
    t0 = time.perf_counter(); ctx.prefill(toks, 0); t1 = time.perf_counter()
    for p, t in enumerate(toks[:32]): ctx.forward(int(t), p)
    t2 = time.perf_counter()
    seq = (t2 - t1) / min(n, 32) * n
    bpt = configs.algorithmic_bytes_per_token(hdr, 0)
    print("%s n=%3d  prefill %.2f ms (%.0f tok/s, weights streamed at %.2f TB/s-equivalent per chunk)  token-by-token %.2f ms  speedup %.1fx"
          % (name, n, 1e3 * (t1 - t0), n / (t1 - t0), bpt * ((n + 63) // 64) / (t1 - t0) / 1e12, 1e3 * seq, seq / (t1 - t0)))
