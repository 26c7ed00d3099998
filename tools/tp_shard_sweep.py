"""Per-phase GEMV time of ONE rank's shard of Llama-2-7B (shard-timing context, L2_TP_SOLO_ID) over launch geometries (development
switches L2_TUNE_NWAVES / L2_TUNE_GRIDCAP / L2_SMALL_MAX behind L2_TEST_HOOKS=1): what should pick_geo choose for 8-45 MB matrices?
  python tools/tp_shard_sweep.py [G ...]"""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["L2_TEST_HOOKS"] = "1"
from llama2_ts_amd import configs, runtime
hdr = configs.header("llama2_7b_L2")
Gs = [int(a) for a in sys.argv[1:]] or [8, 4, 2]
kinds = (("qkv", runtime.T_WQ), ("wo", runtime.T_WO), ("w13", runtime.T_W1), ("w2", runtime.T_W2), ("wcls", runtime.T_WCLS))
for G in Gs:
    print("G = %d" % G)
    for env in ({}, {"L2_TUNE_NWAVES": "1"}, {"L2_TUNE_NWAVES": "2"}, {"L2_TUNE_NWAVES": "4"}, {"L2_SMALL_MAX": "0"}, {"L2_SMALL_MAX": "0", "L2_TUNE_NWAVES": "1"},
                {"L2_TUNE_GRIDCAP": "1024"}, {"L2_TUNE_GRIDCAP": "1024", "L2_TUNE_NWAVES": "2"}, {"L2_TUNE_GRIDCAP": "256", "L2_TUNE_NWAVES": "4"}):
        for k in ("L2_TUNE_NWAVES", "L2_TUNE_GRIDCAP", "L2_SMALL_MAX"):
            os.environ.pop(k, None)
        os.environ.update(env)
        c = runtime.Context(hdr, tp_rank=0, tp_size=G, nccl_id=runtime.TP_SOLO_ID)
        c.synth_fill(1)
        c.decode_greedy(1, 0, 4)
        row = []
        for nm, kind in kinds:
            c.bench_gemv(kind, 0, 20)
            row.append("%s %6.2f" % (nm, c.bench_gemv(kind, 0, 100) * 1e3))
        ms = c.bench_decode(1, 0, 64) / 64
        print("  %-52s %s   step %.4f ms (2 layers)" % (str(env), "  ".join(row), ms))
        c.close()
