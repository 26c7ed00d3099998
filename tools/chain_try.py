"""First contact with the chain launch: tiny shape, eager, compare with the goldens; prints timing."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from llama2_ts_amd import configs, runtime
name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
meta = json.load(open("tests/golden/%s.json" % name)); g = np.load("tests/golden/%s.npz" % name)
ctx = runtime.Context(meta["header"]); ctx.synth_fill(meta["seed"])
ctx.set_option(runtime.OPT_MEGAKERNEL, 1)
keep = {p: i for i, p in enumerate(meta["logit_positions"])}
worst = 0.0
t0 = time.time()
for pos, tok in enumerate(meta["tokens_fed"][:steps]):
    got = np.array(ctx.forward(tok, pos), copy=True)
    if pos in keep:
        worst = max(worst, float(np.abs(got - g["logits"][keep[pos]]).max()))
    assert int(np.argmax(got)) == meta["argmax"][pos], ("argmax", pos)
print(name, "chain forward ok:", steps, "steps, max|dlogit| %.3g, %.1f ms" % (worst, 1e3 * (time.time() - t0)))
toks = ctx.decode_greedy(1, 0, steps)
assert toks.tolist() == meta["argmax"][:steps], "greedy chain mismatch"
print("chain greedy ok")
