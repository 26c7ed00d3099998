"""world_size-2 test of the tensor-parallel algorithm over torch.distributed `gloo`.

Each rank holds only its slices (llama2_ts_amd.tp.tensor_slice) of the oracle generator's tensors, computes
its heads / FFN rows, exchanges fp64 partials with all_reduce and logits slices with all_gather -- the same
collectives, in the same places, that the library issues through RCCL -- and must reproduce the 1-rank
oracle logits.  CPU only; the GPU shard layout itself is checked in test_tp_gpu.py.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _rank_main(rank, world, port, hdr, seed, tokens, out_q):
    import torch.distributed as dist
    import torch
    import oracle_lib as O
    from llama2_ts_amd import tp
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d, h, L, H, _kv, V, S = hdr
    V = abs(V)
    hs = d // H
    orc = O.Oracle(hdr, seed)                       # full tensors; each rank only ever reads its slices

    def sl(kind, layer):
        rows, cols, r0, c0 = tp.tensor_slice(hdr, kind, rank, world)
        full_cols = {2: d, 3: d, 4: d, 5: d, 7: d, 8: h, 9: d, 13: d, 0: d}[kind]
        w = orc.weights(kind, layer).reshape(-1, full_cols)
        return np.ascontiguousarray(w[r0:r0 + rows, c0:c0 + cols])

    f64 = np.float64
    dl, hl, Vl = d // world, h // world, V // world
    kc = np.zeros((L, S, dl), np.float32)
    vc = np.zeros((L, S, dl), np.float32)
    fr = orc.weights(11).reshape(S, hs // 2)
    fi = orc.weights(12).reshape(S, hs // 2)
    emb = orc.weights(0).reshape(V, d)
    results = []

    def rmsnorm(x, w):
        ss = 1.0 / np.sqrt(1e-5 + (x.astype(f64) ** 2).sum() / x.size)
        return (w.astype(f64) * (ss * x.astype(f64))).astype(np.float32)

    for pos, tok in enumerate(tokens):
        x = emb[tok].copy()
        for l in range(L):
            xb = rmsnorm(x, orc.weights(1, l))
            q = (sl(2, l).astype(f64) @ xb.astype(f64)).astype(np.float32)
            k = (sl(3, l).astype(f64) @ xb.astype(f64)).astype(np.float32)
            v = (sl(4, l).astype(f64) @ xb.astype(f64)).astype(np.float32)
            for vec in (q, k):                          # RoPE on adjacent pairs, head-local angles
                a, b = vec[0::2].astype(f64), vec[1::2].astype(f64)
                idx = (np.arange(0, dl, 2) % hs) // 2
                cr, ci = fr[pos, idx].astype(f64), fi[pos, idx].astype(f64)
                vec[0::2] = (a * cr - b * ci).astype(np.float32)
                vec[1::2] = (a * ci + b * cr).astype(np.float32)
            kc[l, pos], vc[l, pos] = k, v
            att_out = np.zeros(dl, np.float32)
            for hh in range(H // world):
                qs = q[hh * hs:(hh + 1) * hs].astype(f64)
                sc = ((kc[l, :pos + 1, hh * hs:(hh + 1) * hs].astype(f64) @ qs) / np.sqrt(f64(hs))).astype(np.float32)
                e = np.exp(sc.astype(f64) - f64(sc.max())).astype(np.float32)
                p = (e.astype(f64) / e.astype(f64).sum()).astype(np.float32)
                att_out[hh * hs:(hh + 1) * hs] = (p.astype(f64) @ vc[l, :pos + 1, hh * hs:(hh + 1) * hs].astype(f64)).astype(np.float32)
            part = torch.from_numpy(sl(5, l).astype(f64) @ att_out.astype(f64))     # fp64 partial of wo . xb
            dist.all_reduce(part, op=dist.ReduceOp.SUM)
            x = x + part.numpy().astype(np.float32)                                  # one rounding, then accum
            xb = rmsnorm(x, orc.weights(6, l))
            h1 = (sl(7, l).astype(f64) @ xb.astype(f64)).astype(np.float32)
            h3 = (sl(9, l).astype(f64) @ xb.astype(f64)).astype(np.float32)
            s1 = (h1.astype(f64) * (1.0 / (1.0 + np.exp(-h1.astype(f64))))).astype(np.float32)
            hb = (s1.astype(f64) * h3.astype(f64)).astype(np.float32)
            part = torch.from_numpy(sl(8, l).astype(f64) @ hb.astype(f64))
            dist.all_reduce(part, op=dist.ReduceOp.SUM)
            x = x + part.numpy().astype(np.float32)
        xn = rmsnorm(x, orc.weights(10))
        wc = orc.weights(13 if hdr[5] < 0 else 0).reshape(V, d)[rank * Vl:(rank + 1) * Vl]
        mine = torch.from_numpy((wc.astype(f64) @ xn.astype(f64)).astype(np.float32))
        parts = [torch.zeros(Vl, dtype=torch.float32) for _ in range(world)]
        dist.all_gather(parts, mine)
        results.append(torch.cat(parts).numpy())
    if rank == 0:
        out_q.put(np.stack(results))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["tiny"])
def test_tp2_over_gloo_matches_single_rank_oracle(name):
    import json
    import torch.multiprocessing as mp
    import oracle_lib as O
    from llama2_ts_amd import tp
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", name + ".json")))
    hdr = tuple(meta["header"])
    assert tp.shards(hdr, 2) and not tp.shards((288, 768, 6, 6, 6, 32000, 256), 4)
    tokens = meta["tokens_fed"][:6]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, hdr, meta["seed"], tokens, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    for pos in range(len(tokens)):
        want = g["logits"][pos]                      # logits of the TRUE reference
        assert np.abs(got[pos] - want).max() <= 1e-5
        assert int(np.argmax(got[pos])) == meta["argmax"][pos]
    orc = O.Oracle(hdr, meta["seed"])                 # and the oracle's own TP restatement agrees
    for pos, tok in enumerate(tokens):
        assert np.abs(orc.forward_tp(tok, pos, 2) - got[pos]).max() <= 1e-5


def test_the_librarys_shard_slices_are_the_plan_the_gloo_run_uses():
    """tests above run a numpy restatement of the partition (llama2_ts_amd/tp.py: tensor_slice) over gloo; the LIBRARY cuts its shards with
    tensor_slice() of csrc/ctx.hip.h.  The two must be the same function: every tensor kind, every rank of 2 / 4 / 8 groups of the shapes
    that shard (Llama-2-7B, its 2-layer twin, tiny, stories15M at 2), through the library's pure debug hook -- no GPU needed."""
    import ctypes as C
    import __graft_entry__ as graft
    from llama2_ts_amd import configs, runtime, tp
    graft.build()
    L = runtime.lib()
    L.l2_debug_tensor_slice.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    checked = 0
    for name, Gs in (("llama2_7b", (2, 4, 8)), ("llama2_7b_L2", (2, 4, 8)), ("tiny", (2, 4)), ("stories15M", (2,))):
        hdr = configs.header(name)
        for G in Gs:
            assert tp.shards(hdr, G)
            for rank in range(G):
                for kind in range(14):
                    out = (C.c_longlong * 6)()
                    assert L.l2_debug_tensor_slice((C.c_int32 * 7)(*hdr), 0, kind, rank, G, out) == 0
                    rows, cols, r0, c0 = tp.tensor_slice(hdr, kind, rank, G)
                    assert (out[0], out[1], out[4], out[5]) == (rows, cols, r0, c0), (name, G, rank, kind, list(out))
                    checked += 1
    assert checked == (3 + 3 + 2 + 1) * 0 + sum(G * 14 for Gs in ((2, 4, 8), (2, 4, 8), (2, 4), (2,)) for G in Gs)
    # the column-sharded matrices cover the full width exactly once across the ranks
    hdr = configs.header("llama2_7b")
    for kind, full in ((5, hdr[0]), (8, hdr[1])):
        spans = sorted((tp.tensor_slice(hdr, kind, r, 8)[3], tp.tensor_slice(hdr, kind, r, 8)[1]) for r in range(8))
        assert spans[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(spans, spans[1:])) and spans[-1][0] + spans[-1][1] == full
