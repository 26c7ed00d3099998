"""Grouped-query shapes and the multi-head model each one is equivalent to.  Shared by tests/test_gqa.py
and oracle/make_goldens.py: the expanded multi-head checkpoint is a file the TRUE reference can run (llama2.ts:117-118 reads
wk / wv as (d, d) whatever n_kv_heads says), which is how the grouped-query path gets a reference-held pin."""
import struct

import numpy as np

import oracle_lib as O

GQA_SHAPES = {
    "tiny_gqa": (64, 176, 2, 4, 2, 512, 64),          # head_size 16, two query heads per cache head
    "wide_gqa": (256, 704, 2, 4, 1, -512, 320),        # head_size 64 (multi-query), unshared classifier, 320 positions
}
GQA_SEEDS = {"tiny_gqa": 9, "wide_gqa": 3}             # seeds of the reference-run fixtures (tests/golden/*_gqa*.json)


def gqa_tensors(hdr, seed, runc_rope=False):
    """{kind: flat float32 array} of the GROUPED-QUERY model (hdr, seed) from the oracle's generator; with `runc_rope` the
    RoPE tables are the ones llama2.c's run.c would compute (what L2_F_GENERATE_ROPE generates) instead of the generator's."""
    O.set_gqa(1)
    try:
        g = O.Oracle(hdr, seed)
        tensors = {}
        for kind in range(14):
            if kind == 13 and hdr[5] > 0:
                continue
            tensors[kind] = np.array(g.weights(kind), copy=True)
        g.close()
    finally:
        O.set_gqa(0)
    if runc_rope:
        tensors[11], tensors[12] = O.rope_runc(hdr)
    return tensors


def expand_to_mha(hdr, tensors):
    """The same tensors with the rows of every cache head of wk / wv repeated for each query head of its group."""
    d, h, L, H, KVH, V, S = hdr
    hs, mul = d // H, H // KVH
    out = dict(tensors)
    for kind in (3, 4):
        w = tensors[kind].reshape(L, KVH, hs, d)
        out[kind] = np.repeat(w, mul, axis=1).reshape(-1)
    return out


def write_v0(path, hdr7, tensors):
    with open(path, "wb") as f:
        f.write(struct.pack("<7i", *hdr7))
        for kind in range(14):
            if kind in tensors:
                f.write(np.ascontiguousarray(tensors[kind], dtype="<f4").tobytes())


def expanded_mha_file(hdr, seed, path, runc_rope=False):
    """v0 checkpoint of the multi-head model equivalent to the grouped-query model (hdr, seed); header n_kv_heads = n_heads."""
    d, h, L, H, KVH, V, S = hdr
    t = expand_to_mha(hdr, gqa_tensors(hdr, seed, runc_rope))
    write_v0(path, (d, h, L, H, H, V, S), t)
    return t


def write_v1(path, hdr, tensors):
    """llama2.c version-1 export: magic "ak42", version 1, the 7 ints, shared-classifier byte, padded to 256 bytes; norms
    first, no freq_cis, wk / wv with n_kv_heads * head_size rows."""
    d, h, L, H, KVH, V, S = hdr
    with open(path, "wb") as f:
        head = struct.pack("<Ii7iB", 0x616b3432, 1, d, h, L, H, KVH, abs(V), S, 1 if V > 0 else 0)
        f.write(head + b"\0" * (256 - len(head)))
        for kind in (1, 6, 10, 0, 2, 3, 4, 5, 7, 8, 9) + (() if V > 0 else (13,)):
            f.write(np.ascontiguousarray(tensors[kind], dtype="<f4").tobytes())
