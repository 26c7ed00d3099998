"""The reference's `argmax` (llama2.ts:364-366) on the device, at its edges: every greedy path of the library -- the blocking call with the
pick on the host, the device loop on the library's own queue (the pick folded into the next token's first launch), the same loop as
replayed hipGraphs and as eager launches, l2_decode_sample at temperature 0, the one-wave pick of the scalar kernels and of a
tensor-parallel rank (maximum over the gathered logits) -- must choose what the reference chooses on logits with EXACT ties of the
maximum in different workgroups and on different key lines, -0 beside +0, +-inf, NaN, nothing but NaN, and NaN at index 0.
The models are tests/argmax_cases.py; the picks asserted are the REAL reference's (tests/golden/argmax_*.json, written by
oracle/make_goldens.py from runs of /root/reference/llama2.ts), which the oracle reproduces (tests/test_oracle_golden.py)."""
import json
import os
import threading

import numpy as np
import pytest

import argmax_cases as A
import oracle_lib as O
from llama2_ts_amd import runtime

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
STEPS = 16


def _gold(case, shape):
    return json.load(open(os.path.join(GOLD, "argmax_%s_%s.json" % (case, shape))))


def _upload(ctx, tensors):
    for kind, layers, count in runtime.tensor_shapes(ctx.cfg):
        per = tensors[kind].reshape(max(layers, 1), -1)
        for layer in range(max(layers, 1)):
            ctx.upload(kind, layer if layers else -1, per[layer])


def _close_enough(got, want):
    """NaN where the reference has NaN, the same infinity where it has one, within 1e-4 elsewhere; zeros keep their sign."""
    got, want = np.asarray(got), np.asarray(want)
    nan = np.isnan(want)
    if not np.array_equal(np.isnan(got), nan):
        return False
    inf = np.isinf(want)
    if not np.array_equal(got[inf], want[inf]) or np.isinf(got[~nan & ~inf]).any():
        return False
    fin = ~nan & ~inf
    zero = fin & (want == 0)
    return bool(np.abs(got[fin] - want[fin]).max(initial=0.0) <= 1e-4 and np.array_equal(np.signbit(got[zero]), np.signbit(want[zero])))


@pytest.mark.parametrize("shape", ["vec", "odd"])
@pytest.mark.parametrize("case", A.CASES)
def test_every_greedy_path_picks_what_the_reference_picks(case, shape):
    meta = _gold(case, shape)
    fed, picks = meta["tokens_fed"], meta["picks"]                   # the reference's own: picks[i] = what it fed at step i + 1
    n = len(picks)
    _, _, want_logits = A.oracle_run(case, shape, STEPS)
    tensors = A.tensors_of(case, shape)
    ctx = runtime.Context(A.SHAPES[shape])
    _upload(ctx, tensors)
    # (1) the drop-in call (llama2.ts:468) + the host's first maximum (llama2.ts:478)
    for pos in range(n):
        lg = ctx.forward(fed[pos], pos)
        assert _close_enough(lg, want_logits[pos]), (case, shape, pos)
        assert runtime.argmax(lg) == picks[pos] == O.argmax(lg), (case, shape, pos)
    # (2) the device loop as it ships (the library's own queue: the pick folded into the next token's first launch; the scalar shape:
    # the one-wave finish), (3) replayed hipGraphs, (4) eager launches, (5) the sampled entry point at temperature 0
    queue = ctx.get_option(runtime.OPT_AQL_QUEUE)
    assert ctx.decode_greedy(1, 0, n).tolist() == picks, (case, shape, "device loop, queue=%d" % queue)
    ctx.set_option(runtime.OPT_AQL_QUEUE, 0)
    assert ctx.decode_greedy(1, 0, n).tolist() == picks, (case, shape, "hipGraph replay")
    ctx.set_option(runtime.OPT_USE_GRAPH, 0)
    assert ctx.decode_greedy(1, 0, n).tolist() == picks, (case, shape, "eager launches")
    ctx.set_option(runtime.OPT_USE_GRAPH, 1)
    ctx.set_option(runtime.OPT_AQL_QUEUE, 1)
    toks, rng = ctx.decode_sample(1, 0, n, 0.0, 0.9, 1234)
    assert toks.tolist() == picks and rng == 1234, (case, shape, "l2_decode_sample at temperature 0 (no draw)")
    # a run that starts in the middle of the sequence: the first token comes from the host, the following ones from the keys
    assert ctx.decode_greedy(fed[5], 5, n - 5).tolist() == picks[5:], (case, shape, "device loop from position 5")
    ctx.close()


@pytest.mark.parametrize("collective", ["p2p", "rccl"])
@pytest.mark.parametrize("case", A.CASES)
def test_tensor_parallel_pick_over_gathered_logits(case, collective):
    """A tensor-parallel rank takes the maximum over the GATHERED logits with one workgroup (argmax_advance_kernel): a 1-rank
    communicator (RCCL collectives / the one-shot exchange kernels) and a 2-rank loopback group whose slices split the tied rows."""
    meta = _gold(case, "vec")
    picks = meta["picks"]
    tensors = A.tensors_of(case, "vec")
    os.environ["L2_TP_FORCE_COMM"] = "1"
    os.environ["L2_TP_ALLREDUCE"] = collective
    try:
        ctx = runtime.Context(A.SHAPES["vec"])
        assert ctx.tp_mode_id() in (1, 2, 3)
        _upload(ctx, tensors)
        assert ctx.decode_greedy(1, 0, len(picks)).tolist() == picks, (case, collective, "1-rank communicator")
        ctx.close()
    finally:
        del os.environ["L2_TP_FORCE_COMM"]
    G, gid = 2, bytes([2, 77] + [len(case)] * 126)
    out, errs = [None] * G, [None] * G

    def rank_main(r):
        try:
            c = runtime.Context(A.SHAPES["vec"], tp_rank=r, tp_size=G, nccl_id=gid)
            _upload(c, tensors)
            out[r] = c.decode_greedy(1, 0, len(picks)).tolist()
            c.close()
        except BaseException as e:      # surfaced by the main thread
            errs[r] = e

    os.environ["L2_TP_LOOPBACK"] = "1"
    try:
        ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(G)]
        [t.start() for t in ts]
        [t.join(120) for t in ts]
        assert not any(t.is_alive() for t in ts), "a rank hung"
    finally:
        del os.environ["L2_TP_LOOPBACK"]
        del os.environ["L2_TP_ALLREDUCE"]
    for e in errs:
        if e is not None:
            raise e
    assert out[0] == picks and out[1] == picks, (case, collective, "2-rank loopback group")
