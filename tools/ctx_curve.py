"""ms/token of the device-resident decode loop in windows along the context (attention split levels)."""
import os as _os; _os.environ.setdefault("L2_TEST_HOOKS", "1")   # development switches are gated
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llama2_ts_amd import configs, runtime
name = sys.argv[1] if len(sys.argv) > 1 else "llama2_7b"
hdr = configs.header(name)
ctx = runtime.Context(hdr); ctx.synth_fill(1)
ctx.bench_decode(1, 0, 8)
S = hdr[6]
out = []
for p0 in (0, 64, 128, 192, 256, 320, 448, 512, 768, 960, 1024, 1088, 1536, 1920):
    if p0 + 64 > S: break
    ctx.bench_decode(1, p0, 4)
    ms = ctx.bench_decode(1, p0, 64) / 64
    bpt = sum(configs.algorithmic_bytes_per_token(hdr, p) for p in range(p0, p0 + 64)) / 64
    out.append("%d:%.3f(%.0f%%)" % (p0, ms, 100 * bpt / (ms * 1e-3) / 8e12))
print(name, "splits", os.environ.get("L2_ATTN_SPLITS", "auto"), " ".join(out))
