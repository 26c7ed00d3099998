"""What tests/test_coherence_gpu.py measures, as a script of its own so that it can also run in a fresh process against ANOTHER build of the
library (L2_LIB_PATH): per shape, a decode with the coherence rule's adversary behind every launch (L2_DEBUG_POLLUTE=1: every CU pulls every
mutable line of the step into its L1 with plain loads, csrc/kernels.hip.h: l1_pollute_kernel) on the library's own queue -- where no launch
but a token's first acquires -- against replayed hipGraphs without the adversary (every node acquires and releases).  Prints one JSON line:
{shape: {"queue": 0|1, "tokens_equal": bool, "logits_equal": bool, "sampled_equal": bool}}.

  python tests/coherence_probe.py            (needs L2_TEST_HOOKS=1 in the environment: the polluter is a test hook)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

from llama2_ts_amd import runtime  # noqa: E402

# name: (header, greedy steps, positions fed through the blocking call)
SHAPES = {
    "tiny": ((64, 176, 2, 4, 4, 512, 64), 64, 24),                                   # latency form, one workgroup per head
    "fused": ((288, 768, 2, 6, 6, 2000, 320), 320, 40),                              # stories15M width: two launches -> the fused QKV + attention launch -> 8 splits per head
    "wide110": ((768, 2048, 2, 12, 12, 4000, 300), 300, 160),                        # stories110M width: fused from 129 rows, split beyond 256
    "stream7b": ((4096, 11008, 1, 32, 32, -3200, 176), 176, 24),                     # 7B width: the streaming form on repacked matrices, split attention beyond 144 rows
    "odd": ((66, 170, 2, 3, 3, -259, 33), 33, 16),                                   # n % 4 != 0: the scalar kernels and the one-wave pick
}
SEED = 5


def run(shape, pollute, queue):
    hdr, steps, nfwd = SHAPES[shape]
    if pollute:
        os.environ["L2_DEBUG_POLLUTE"] = "1"
    else:
        os.environ.pop("L2_DEBUG_POLLUTE", None)
    ctx = runtime.Context(hdr)
    ctx.synth_fill(SEED)
    ctx.set_option(runtime.OPT_AQL_QUEUE, queue)
    toks = ctx.decode_greedy(1, 0, steps)
    used = ctx.get_option(runtime.OPT_AQL_QUEUE)
    logits = []
    for pos in range(nfwd):      # the blocking call, fed the greedy tokens: positions 0 .. nfwd - 1 again (the caches are rewritten row by row)
        logits.append(np.array(ctx.forward(1 if pos == 0 else int(toks[pos - 1]), pos), copy=True))
    samp, rng = ctx.decode_sample(1, 0, min(steps, 48), 0.9, 0.9, 77)
    ctx.close()
    os.environ.pop("L2_DEBUG_POLLUTE", None)
    return toks, np.stack(logits), samp, rng, used


def probe(shapes=None):
    out = {}
    for shape in (shapes or SHAPES):
        ref = run(shape, pollute=False, queue=0)
        got = run(shape, pollute=True, queue=1)
        out[shape] = {"queue": int(got[4]), "tokens_equal": bool(np.array_equal(ref[0], got[0])),
                      "logits_equal": bool(np.array_equal(ref[1].view(np.uint32), got[1].view(np.uint32))),
                      "sampled_equal": bool(np.array_equal(ref[2], got[2]) and ref[3] == got[3])}
    return out


if __name__ == "__main__":
    print(json.dumps(probe(sys.argv[1:] or None)))
