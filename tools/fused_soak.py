"""Soak of the fused QKV + attention launch: the same 256-token greedy decode over and over (device loop, one hipGraph replay per token);
every run must reproduce the reference's golden tokens -- a hand-off granule taken too early, a tag that repeats or a wait that gives up
would show as a different token or an error.  python tools/fused_soak.py <config> <runs>   (L2_FUSE_MIN_ROWS=0 behind L2_TEST_HOOKS=1
fuses from the first row on)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llama2_ts_amd import configs, runtime
name = sys.argv[1]; runs = int(sys.argv[2])
meta = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", name + ".json")))
n = min(256, len(meta["argmax"]))
ctx = runtime.Context(meta["header"]); ctx.synth_fill(meta["seed"])
bad = 0; t0 = time.time()
for r in range(runs):
    toks = ctx.decode_greedy(1, 0, n).tolist()
    if toks != meta["argmax"][:n]:
        bad += 1
        first = next(i for i in range(n) if toks[i] != meta["argmax"][i])
        print("run %d differs from the reference at step %d" % (r, first))
print("%s: %d runs x %d tokens, %d differing runs, %.1f s (%s)" % (name, runs, n, bad, time.time() - t0, os.environ.get("L2_FUSE_MIN_ROWS", "default policy")))
ctx.close()
sys.exit(1 if bad else 0)
