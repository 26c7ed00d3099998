import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from llama2_ts_amd import configs, runtime
name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
meta = json.load(open("tests/golden/%s.json" % name)); g = np.load("tests/golden/%s.npz" % name)
res = {}
for chain in (0, 1):
    ctx = runtime.Context(meta["header"]); ctx.synth_fill(meta["seed"])
    ctx.set_option(runtime.OPT_MEGAKERNEL, chain)
    lg = np.array(ctx.forward(1, 0), copy=True)
    res[chain] = {n: ctx.read_state(n) for n in ("x", "xb", "xb2", "hb", "hb2", "q", "k", "v", "att")}
    res[chain]["logits"] = lg
    ctx.close()
for n in res[0]:
    d = np.abs(res[0][n] - res[1][n]).max()
    gd = np.abs(res[0][n] - g[n][0]).max() if n in g.files and n != "logits" else -1
    print("%-7s chain-vs-plain max diff %.3g   plain-vs-golden %.3g   nan=%s" % (n, d, gd, np.isnan(res[1][n]).any()))
