"""Sweep launch geometry of the GEMV phase kernels (env L2_TUNE_*) and print GB/s per matrix kind."""
import os as _os; _os.environ.setdefault("L2_TEST_HOOKS", "1")   # development switches are gated
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llama2_ts_amd import configs, runtime

name = sys.argv[1] if len(sys.argv) > 1 else "llama2_7b_L2"
hdr = configs.header(name)
cfg = runtime.Config(hdr)
d, h, V = cfg.dim, cfg.hidden_dim, cfg.vocab_size
nbytes = {"qkv": 3 * d * d * 4, "wo": d * d * 4, "w13": 2 * d * h * 4, "w2": d * h * 4, "wcls": V * d * 4}
kinds = {"qkv": runtime.T_WQ, "wo": runtime.T_WO, "w13": runtime.T_W1, "w2": runtime.T_W2, "wcls": runtime.T_WCLS}
Rs = [int(x) for x in os.environ.get("SWEEP_R", "0").split(",")]
Us = [int(x) for x in os.environ.get("SWEEP_U", "0").split(",")]
NWs = [int(x) for x in os.environ.get("SWEEP_NW", "0").split(",")]
CAPs = [int(x) for x in os.environ.get("SWEEP_CAP", "0").split(",")]
iters = int(os.environ.get("SWEEP_ITERS", "50"))
print("config", name, hdr)
for R, U, nw, cap in itertools.product(Rs, Us, NWs, CAPs):
    os.environ["L2_TUNE_R"] = str(R); os.environ["L2_TUNE_U"] = str(U)
    os.environ["L2_TUNE_NWAVES"] = str(nw); os.environ["L2_TUNE_GRIDCAP"] = str(cap)
    ctx = runtime.Context(hdr); ctx.synth_fill(1)
    ctx.forward(1, 0)
    row = []
    for nm, kind in kinds.items():
        ms = min(ctx.bench_gemv(kind, 1 if cfg.n_layers > 1 else 0, iters) for _ in range(3))
        row.append("%s %7.1fus %6.0fGB/s" % (nm, ms * 1e3, nbytes[nm] / ms / 1e6))
    print("R=%d U=%d nw=%d cap=%4d | " % (R, U, nw, cap) + " | ".join(row), flush=True)
    ctx.close()
