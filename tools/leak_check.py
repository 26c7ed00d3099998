"""Device memory before / after many create-use-destroy cycles, per feature (leak check)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from llama2_ts_amd import configs, runtime
hdr = configs.header(sys.argv[1] if len(sys.argv) > 1 else "stories15M")
def cycle(what):
    ctx = runtime.Context(hdr); ctx.synth_fill(1)
    if "forward" in what: ctx.forward(1, 0)
    if "greedy" in what: ctx.decode_greedy(1, 0, 8)
    if "sample" in what: ctx.decode_sample(1, 0, 8, 0.9, 0.9, 3)
    if "prefill" in what: ctx.prefill(np.arange(2, 2 + min(200, hdr[6] - 1), dtype=np.int32), 0)
    ctx.close()
cycle("forward greedy sample prefill")          # one-time allocations of the runtime itself
for what in ("create", "forward", "greedy", "sample", "prefill"):
    torch.cuda.synchronize(); f0 = torch.cuda.mem_get_info()[0]
    for _ in range(40): cycle(what)
    torch.cuda.synchronize(); f1 = torch.cuda.mem_get_info()[0]
    print("%-8s x40: %+.2f MB" % (what, (f0 - f1) / 1e6))
