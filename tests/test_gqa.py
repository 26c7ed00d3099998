"""grouped-query checkpoints and checkpoints without RoPE tables.

llama2.ts parses n_kv_heads and ignores it (llama2.ts:86, 117-118) and reads freq_cis from the file (:125-126): it cannot
run such checkpoints directly.  It CAN run the multi-head expansion of a grouped-query model (wk / wv rows of each cache head
repeated for its query heads) and a v0 file whose freq_cis_* hold the tables a version-1 export implies -- oracle/make_goldens.py
recorded both runs (tests/golden/wide_gqa_mha, tiny_gqa_rope_mha); the first group of tests below compares the oracle (bit for
bit) and the HIP path (<= 1e-4, argmax exact) with those reference outputs.  The older, transitive checks remain:
  * the oracle's grouped-query switch (oracle/llama2_oracle.c: orc_set_gqa) changes NOTHING when n_kv_heads == n_heads,
    and a grouped-query model equals -- bit for bit -- the multi-head model whose wk / wv rows repeat each cache head
    for its group of query heads; that multi-head model runs on the restatement that IS pinned to the reference;
  * the HIP path (l2_create_ex with L2_F_GQA) is then compared with that oracle, with the same expanded multi-head
    model on the ordinary HIP path, and through the version-1 checkpoint loader.
"""
import hashlib
import json
import os
import struct

import numpy as np
import pytest

import oracle_lib as O

from gqa_cases import GQA_SEEDS, GQA_SHAPES, expanded_mha_file, gqa_tensors, write_v0, write_v1

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def load_gold(name):
    return json.load(open(os.path.join(GOLD, name + ".json"))), np.load(os.path.join(GOLD, name + ".npz"))


# ---- the reference-held pin: the TRUE reference ran the multi-head expansion (oracle/make_goldens.py, *_mha fixtures) ----------

@pytest.mark.parametrize("name", ["wide_gqa_mha", "tiny_gqa_rope_mha"])
def test_oracle_gqa_is_bit_identical_to_the_reference_run_of_the_expanded_model(name, tmp_path):
    """The grouped-query restatement (orc_set_gqa) on the GROUPED-QUERY tensors reproduces, bit for bit and step by step, what
    the real reference computed on the expanded multi-head file (llama2.ts:117-118 reads wk / wv as (d, d)); for `_rope` the
    file's freq_cis_* are llama2.c run.c's tables (llama2.ts:125-126 reads whatever the file holds)."""
    meta, g = load_gold(name)
    hdr = tuple(meta["gqa_header"])
    p = str(tmp_path / "gqa.bin")
    write_v0(p, hdr, gqa_tensors(hdr, meta["seed"], runc_rope="_rope" in name))
    O.set_gqa(1)
    try:
        orc = O.Oracle(hdr, path=p)
        for pos, tok in enumerate(meta["tokens_fed"]):
            lg = orc.forward(tok, pos)
            assert hashlib.sha256(lg.tobytes()).hexdigest() == meta["logits_sha256"][pos], (name, pos)
        orc.close()
    finally:
        O.set_gqa(0)


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{}, {"L2_ATTN_SPLITS": "4"}, {"L2_SMALL_MAX": "0"}])
def test_hip_gqa_matches_the_reference_run_of_the_expanded_model(env, monkeypatch):
    """L2_F_GQA on the grouped-query tensors vs the TRUE reference's logits / argmax for the expanded model: every step of the
    320-position context (both attention split levels), logits <= 1e-4 at the kept positions, then the device greedy loop."""
    from llama2_ts_amd import runtime
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    meta, g = load_gold("wide_gqa_mha")
    hdr = tuple(meta["gqa_header"])
    t = gqa_tensors(hdr, meta["seed"])
    ctx = runtime.Context(hdr, flags=runtime.F_GQA)
    for kind, layers, count in runtime.tensor_shapes(ctx.cfg, gqa=True):
        per = t[kind].size // max(layers, 1)
        assert per == count
        for layer in range(max(layers, 1)):
            ctx.upload(kind, layer if layers else -1, t[kind][layer * per:(layer + 1) * per])
    keep = {p: i for i, p in enumerate(meta["logit_positions"])}
    worst = 0.0
    for pos, tok in enumerate(meta["tokens_fed"]):
        got = ctx.forward(tok, pos)
        assert runtime.argmax(got) == meta["argmax"][pos], pos
        if pos in keep:
            worst = max(worst, float(np.abs(got - g["logits"][keep[pos]]).max()))
    assert worst <= 1e-4, worst
    assert ctx.decode_greedy(1, 0, len(meta["argmax"])).tolist() == meta["argmax"]
    ctx.close()


@pytest.mark.gpu
def test_hip_version_1_checkpoint_matches_the_reference_run_with_run_c_tables(tmp_path):
    """A version-1 export (no freq_cis, grouped wk / wv) through l2_load_checkpoint (L2_F_GQA | L2_F_GENERATE_ROPE) vs the TRUE
    reference on the expanded v0 file whose freq_cis_* hold run.c's tables: every logit of all 64 positions <= 1e-4, argmax
    identical; the generated tables themselves equal the file's (same libm formula) within 1 ulp."""
    from llama2_ts_amd import runtime
    meta, g = load_gold("tiny_gqa_rope_mha")
    hdr = tuple(meta["gqa_header"])
    t = gqa_tensors(hdr, meta["seed"], runc_rope=True)
    p = str(tmp_path / "v1.bin")
    write_v1(p, hdr, t)
    cfg, state, weights, nbytes = runtime.load_checkpoint_native(p)
    assert nbytes == os.path.getsize(p) and cfg.header == hdr
    ctx = weights.ctx
    S, hs2 = hdr[6], (hdr[0] // hdr[3]) // 2
    assert np.abs(ctx.read_tensor(runtime.T_FREQ_REAL, 0, 0, S * hs2) - t[11]).max() <= 1.2e-7
    assert np.abs(ctx.read_tensor(runtime.T_FREQ_IMAG, 0, 0, S * hs2) - t[12]).max() <= 1.2e-7
    for pos, tok in enumerate(meta["tokens_fed"]):
        got = ctx.forward(tok, pos)
        assert runtime.argmax(got) == meta["argmax"][pos], pos
        assert np.abs(got - g["logits"][pos]).max() <= 1e-4, pos
    ctx.close()


@pytest.mark.parametrize("name", sorted(GQA_SHAPES))
def test_oracle_gqa_equals_the_expanded_multi_head_model(name, tmp_path):
    hdr = GQA_SHAPES[name]
    p = str(tmp_path / "mha.bin")
    expanded_mha_file(hdr, 3, p)
    mha = O.Oracle(hdr[:4] + (hdr[3],) + hdr[5:], path=p)      # the reference-pinned multi-head path
    O.set_gqa(1)
    try:
        gqa = O.Oracle(hdr, 3)
        tok = 1
        for pos in range(20):
            a, b = gqa.forward(tok, pos), mha.forward(tok, pos)
            assert np.array_equal(bits(a), bits(b)), pos
            tok = O.argmax(a)
        gqa.close()
    finally:
        O.set_gqa(0)
    mha.close()


def test_oracle_gqa_switch_is_a_no_op_for_multi_head_checkpoints():
    hdr = (64, 176, 2, 4, 4, 512, 64)
    a = O.Oracle(hdr, 5)
    O.set_gqa(1)
    try:
        b = O.Oracle(hdr, 5)
        for pos, tok in enumerate([1, 7, 9, 300]):
            assert np.array_equal(bits(a.forward(tok, pos)), bits(b.forward(tok, pos)))
        b.close()
    finally:
        O.set_gqa(0)
    a.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,env", [("tiny_gqa", {}), ("wide_gqa", {}), ("wide_gqa", {"L2_ATTN_SPLITS": "4"}), ("wide_gqa", {"L2_SMALL_MAX": "0"})])
def test_hip_gqa_matches_the_oracle_and_the_expanded_multi_head_model(name, env, tmp_path, monkeypatch):
    from llama2_ts_amd import runtime
    import __graft_entry__ as graft
    graft.build()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    hdr = GQA_SHAPES[name]
    steps = min(hdr[6], 160)
    p = str(tmp_path / "mha.bin")
    expanded_mha_file(hdr, 3, p)
    _, _, mha_w = runtime.load_checkpoint(p)                   # ordinary (reference-semantics) context, expanded weights
    O.set_gqa(1)
    try:
        orc = O.Oracle(hdr, 3)
        ctx = runtime.Context(hdr, flags=runtime.F_GQA)
        for kind, layers, count in runtime.tensor_shapes(ctx.cfg, gqa=True):
            for layer in range(max(layers, 1)):
                w = orc.weights(kind, layer if layers else -1)
                assert w.size == count
                ctx.upload(kind, layer if layers else -1, w)
        tok, worst = 1, 0.0
        for pos in range(steps):
            got = ctx.forward(tok, pos)
            same = mha_w.ctx.forward(tok, pos)
            assert np.array_equal(bits(got), bits(same)), pos      # identical arithmetic, cache head by cache head
            if pos < 24 or pos in (143, 144, 145, steps - 1):
                want = orc.forward(tok, pos)
                worst = max(worst, float(np.abs(got - want).max()))
                assert runtime.argmax(got) == O.argmax(want), pos
            elif pos < steps:
                orc.forward(tok, pos)
            tok = runtime.argmax(got)
        assert worst <= 1e-4
        d, S, L = ctx.cfg.dim, ctx.cfg.seq_len, ctx.cfg.n_layers
        kvd = ctx.cfg.n_kv_heads * ctx.cfg.head_size
        kc = ctx.read_state("key_cache").reshape(L, S, kvd)[:, :steps]
        assert np.abs(kc - orc.state("key_cache")[:L * S * kvd].reshape(L, S, kvd)[:, :steps]).max() <= 1e-4
        # exact attention mode and the device-resident greedy loop on the same context
        ctx.set_option(runtime.OPT_EXACT_ATTENTION, 1)
        assert np.abs(ctx.forward(1, 0) - mha_w.ctx.forward(1, 0)).max() <= 1e-5
        ctx.set_option(runtime.OPT_EXACT_ATTENTION, 0)
        a, b = ctx.decode_greedy(1, 0, 40), mha_w.ctx.decode_greedy(1, 0, 40)
        assert a.tolist() == b.tolist()
        ctx.close(); orc.close()
    finally:
        O.set_gqa(0)
    mha_w.ctx.close()


@pytest.mark.gpu
def test_version_1_checkpoint_loader_and_generated_rope(tmp_path):
    """A llama2.c version-1 export (magic "ak42", norms first, no freq_cis, wk / wv with n_kv_heads * head_size rows)
    through l2_load_checkpoint == the same tensors uploaded one by one into an l2_create_ex(GQA | GENERATE_ROPE) context;
    the generated tables follow run.c's per-position formula."""
    from llama2_ts_amd import runtime
    hdr = GQA_SHAPES["tiny_gqa"]
    d, h, L, H, KVH, V, S = hdr
    O.set_gqa(1)
    try:
        orc = O.Oracle(hdr, 9)
        t = {k: np.array(orc.weights(k), copy=True) for k in range(11)}
        orc.close()
    finally:
        O.set_gqa(0)
    p = str(tmp_path / "v1.bin")
    with open(p, "wb") as f:
        head = struct.pack("<Ii7iB", 0x616b3432, 1, d, h, L, H, KVH, abs(V), S, 1)
        f.write(head + b"\0" * (256 - len(head)))
        for kind in (1, 6, 10, 0, 2, 3, 4, 5, 7, 8, 9):
            f.write(t[kind].astype("<f4").tobytes())
    cfg, state, weights, nbytes = runtime.load_checkpoint_native(p)
    assert nbytes == os.path.getsize(p) and cfg.header == hdr
    ctx = runtime.Context(hdr, flags=runtime.F_GQA | runtime.F_GENERATE_ROPE)
    for kind, layers, count in runtime.tensor_shapes(ctx.cfg, gqa=True):
        if kind in (runtime.T_FREQ_REAL, runtime.T_FREQ_IMAG):
            continue
        for layer in range(max(layers, 1)):
            n = t[kind].size // max(layers, 1)
            ctx.upload(kind, layer if layers else -1, t[kind][layer * n:(layer + 1) * n])
    hs2 = (d // H) // 2
    pos = np.arange(S, dtype=np.float32)[:, None]
    freq = (np.float32(1.0) / np.power(np.float32(10000.0), (2 * np.arange(hs2, dtype=np.float32)) / np.float32(d // H))).astype(np.float32)
    assert np.abs(ctx.read_tensor(runtime.T_FREQ_REAL, 0, 0, S * hs2).reshape(S, hs2) - np.cos(pos * freq)).max() <= 2e-6
    assert np.abs(ctx.read_tensor(runtime.T_FREQ_IMAG, 0, 0, S * hs2).reshape(S, hs2) - np.sin(pos * freq)).max() <= 2e-6
    tok = 1
    for q in range(16):
        a, b = weights.ctx.forward(tok, q), ctx.forward(tok, q)
        assert np.array_equal(bits(a), bits(b)), q
        tok = runtime.argmax(a)
    ctx.close(); weights.ctx.close()
