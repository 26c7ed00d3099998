"""Tensor-parallel shard layout on one GPU (no communicator): every rank's slices, whether generated on the
device or cut out of full host arrays by l2_upload, must equal llama2_ts_amd.tp.tensor_slice of the oracle's
tensors.  The collectives themselves need >1 GPU and are exercised by bench.py --gpus N on the driver's node;
their arithmetic is covered over gloo in test_tp_gloo.py."""
import os

import numpy as np
import pytest

import oracle_lib as O
from llama2_ts_amd import configs, runtime, tp

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("G", [2, 4])
def test_rank_slices_match_plan(G):
    import __graft_entry__ as graft
    graft.build()
    os.environ["L2_TP_NO_COMM"] = "1"
    try:
        hdr = configs.header("tiny")
        orc = O.Oracle(hdr, 5)
        d, h, L, H, _kv, V, S = hdr
        full_cols = {0: d, 2: d, 3: d, 4: d, 5: d, 7: d, 8: h, 9: d}
        for rank in range(G):
            for mode in ("synth", "upload"):
                ctx = runtime.Context(hdr, tp_rank=rank, tp_size=G, nccl_id=bytes(128))
                if mode == "synth":
                    ctx.synth_fill(5)
                else:
                    for kind, layers, count in runtime.tensor_shapes(ctx.cfg):
                        for layer in range(max(layers, 1)):
                            ctx.upload(kind, layer if layers else -1, orc.weights(kind, layer if layers else -1))
                for kind in (2, 3, 4, 5, 7, 8, 9):
                    rows, cols, r0, c0 = tp.tensor_slice(hdr, kind, rank, G)
                    for layer in (0, L - 1):
                        want = orc.weights(kind, layer).reshape(-1, full_cols[kind])[r0:r0 + rows, c0:c0 + cols]
                        got = ctx.read_tensor(kind, layer, 0, rows * cols).reshape(rows, cols)
                        assert np.array_equal(got, want), (mode, rank, kind, layer)
                # shared classifier: the rank's rows of the (replicated) embedding table
                rows, cols, r0, _ = tp.tensor_slice(hdr, 13, rank, G)
                got = ctx.read_tensor(13, 0, 0, rows * cols).reshape(rows, cols)
                assert np.array_equal(got, orc.weights(0).reshape(V, d)[r0:r0 + rows])
                with pytest.raises(runtime.L2Error) as e:
                    ctx.forward(1, 0)
                assert e.value.code == -6
                ctx.close()
    finally:
        del os.environ["L2_TP_NO_COMM"]


def test_rccl_path_with_a_one_rank_communicator():
    """L2_TP_FORCE_COMM=1: a 1-rank RCCL communicator drives the tensor-parallel code path (fp64 partials,
    ncclAllReduce(double, sum), residual kernel, ncclAllGather of the logits) on a single GPU.  Results must
    match the goldens of the TRUE reference like the ordinary path does."""
    import json
    meta = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tiny.json")))
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "tiny.npz"))
    os.environ["L2_TP_FORCE_COMM"] = "1"
    try:
        ctx = runtime.Context(meta["header"])
        ctx.synth_fill(meta["seed"])
        for pos, tok in enumerate(meta["tokens_fed"][:24]):
            got = np.array(ctx.forward(tok, pos), copy=True)
            assert np.abs(got - g["logits"][pos]).max() <= 1e-4
            assert runtime.argmax(got) == meta["argmax"][pos]
        toks = ctx.decode_greedy(1, 0, 32)
        assert toks.tolist() == meta["argmax"][:32]
        ctx.close()
    finally:
        del os.environ["L2_TP_FORCE_COMM"]
