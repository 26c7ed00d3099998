import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from llama2_ts_amd import configs, runtime
ctx = runtime.Context(configs.header("llama2_7b_L2")); ctx.synth_fill(1)
toks = np.arange(2, 2 + int(os.environ.get("PF_TOKENS", "64")), dtype=np.int32)   # one chunk: 16, 32 or 64 tokens
for _ in range(4): ctx.prefill(toks, 0)
