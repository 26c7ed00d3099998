"""ms per token of a device-resident sampled decode (l2_decode_sample) next to the greedy loop (l2_decode_greedy)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from llama2_ts_amd import configs, runtime
name = sys.argv[1] if len(sys.argv) > 1 else "stories110M"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ctx = runtime.Context(configs.header(name)); ctx.synth_fill(1)
def timed(f):
    f(); t = time.perf_counter(); f(); return (time.perf_counter() - t) * 1e3 / n
g = timed(lambda: ctx.decode_greedy(1, 0, n))
s = timed(lambda: ctx.decode_sample(1, 0, n, 0.9, 1.0, 42))
p = timed(lambda: ctx.decode_sample(1, 0, n, 0.9, 0.9, 42))
print("%s: greedy %.3f ms/token, sample(t=0.9) %.3f (+%.0f us), top-p 0.9 %.3f (+%.0f us)" % (name, g, s, (s - g) * 1e3, p, (p - g) * 1e3))
