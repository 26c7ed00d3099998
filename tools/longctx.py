"""Decode speed at a long context: `steps` tokens starting at pos0 (KV history is whatever the cache holds)."""
import os as _os; _os.environ.setdefault("L2_TEST_HOOKS", "1")   # development switches are gated
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llama2_ts_amd import configs, runtime
name, pos0, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
ctx = runtime.Context(configs.header(name)); ctx.synth_fill(1)
ctx.bench_decode(1, 0, 8)
ms = ctx.bench_decode(1, pos0, steps)
print(name, "splits", os.environ.get("L2_ATTN_SPLITS", "default"), "pos0", pos0, "ms/token %.4f" % (ms / steps))
