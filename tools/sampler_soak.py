"""Soak of the device sampler's default (margin) form against the form with every running sum exact (L2_SAMPLER_CHAIN=1): the same sampled
decodes, thousands of tokens per setting, must give the same token ids and RNG state; prints how many tokens the margin form sent through its
serial loop.  python tools/sampler_soak.py <config> <tokens per setting>"""
import os, sys, time
os.environ.setdefault("L2_TEST_HOOKS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llama2_ts_amd import configs, runtime
name = sys.argv[1] if len(sys.argv) > 1 else "stories15M"
total = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
hdr = configs.header(name)
S = hdr[6]
bad = 0
for temperature, topp in ((0.9, 1.0), (0.9, 0.9), (1.3, 0.5), (0.4, 0.95), (1e6, 0.7), (2.0, 1.0)):
    runs = {}
    for form in ("margin", "chain"):
        os.environ["L2_SAMPLER_CHAIN"] = "1" if form == "chain" else "0"
        ctx = runtime.Context(hdr); ctx.synth_fill(1)
        toks, rng, t0 = [], 12345, time.time()
        tok = 1
        while len(toks) < total:
            n = min(S, total - len(toks))
            t, rng = ctx.decode_sample(1, 0, n, temperature, topp, rng)      # a fresh sequence from BOS, the RNG state carried on
            toks += t.tolist()
        runs[form] = (toks, rng, time.time() - t0, ctx.get_option(runtime.OPT_SAMPLED_SERIAL) if form == "margin" else None)
        ctx.close()
    same = runs["margin"][0] == runs["chain"][0] and runs["margin"][1] == runs["chain"][1]
    bad += 0 if same else 1
    print("%s t=%g p=%g: %d tokens, %s, serial loop %d, %.2f s (margin) / %.2f s (chain)" % (name, temperature, topp, total, "identical" if same else "DIFFERENT", runs["margin"][3], runs["margin"][2], runs["chain"][2]))
sys.exit(1 if bad else 0)
