"""Soak of the decode step on the library's own AQL queue (no acquire fence between the launches of a token: csrc/aql_queue.h, the
coherence rule of kernels.hip.h): whole-context greedy runs over and over -- the device loop, the blocking call, the sampled loop --
every token compared with the real reference's golden (greedy) or with the same run through replayed hipGraphs (sampled).  A byte
read through a stale cache line would show as a different token.
  python tools/aql_soak.py <config> <runs> [steps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from llama2_ts_amd import configs, runtime
name = sys.argv[1]; runs = int(sys.argv[2])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
meta = json.load(open(os.path.join(root, "tests", "golden", name + ".json")))
n = int(sys.argv[3]) if len(sys.argv) > 3 else len(meta["argmax"])
want = meta["argmax"][:n]
ctx = runtime.Context(meta["header"]); ctx.synth_fill(meta["seed"])
ref = runtime.Context(meta["header"]); ref.set_option(runtime.OPT_AQL_QUEUE, 0); ref.synth_fill(meta["seed"])
bad = 0; t0 = time.time(); tokens = 0
for r in range(runs):
    toks = ctx.decode_greedy(1, 0, n).tolist(); tokens += n
    if toks != want:
        bad += 1; print("greedy run %d differs from the reference at step %d" % (r, next(i for i in range(n) if toks[i] != want[i])), flush=True)
    if r % 8 == 0:      # the blocking call over the first 96 positions, then a sampled run, against the graph-replay context
        tok = 1
        for pos in range(min(96, n)):
            lg = ctx.forward(tok, pos, view=True); tokens += 1
            if runtime.argmax(lg) != want[pos]:
                bad += 1; print("blocking call, run %d: argmax differs at position %d" % (r, pos), flush=True); break
            tok = want[pos]
        m = min(n, 200)
        for (t, p, seed) in ((0.9, 1.0, 42 + r), (1.0, 0.9, 7 + r)):
            a, sa = ctx.decode_sample(1, 0, m, t, p, seed); b, sb = ref.decode_sample(1, 0, m, t, p, seed); tokens += m
            if a.tolist() != b.tolist() or sa != sb:
                bad += 1; print("sampled run %d (t=%g p=%g) differs from graph replay" % (r, t, p), flush=True)
assert ctx.get_option(runtime.OPT_AQL_QUEUE) == 1, "the queue was not in use"
print("%s: %d greedy runs x %d tokens (+ blocking calls and sampled runs every 8th): %d tokens, %d differing runs, %.1f s" % (name, runs, n, tokens, bad, time.time() - t0))
ctx.close(); ref.close()
sys.exit(1 if bad else 0)
