"""GPU parity: the HIP path (through the C ABI) against (1) golden vectors recorded from the TRUE
reference and (2) the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): token ids / argmax identical, |logit - reference| <= 1e-4.
The reference accumulates in fp64 and stores fp32; so does the GPU path, hence most logits are
bit-identical -- the tests also report that fraction.
"""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as O
import sum_cases
from llama2_ts_amd import configs, runtime

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-4   # north_star: logits within 1e-4 fp32


def load_gold(name):
    meta = json.load(open(os.path.join(GOLD, name + ".json")))
    return meta, np.load(os.path.join(GOLD, name + ".npz"))


def upload_from_oracle(ctx, orc):
    """Hand the oracle generator's tensors to l2_upload one Float32Array at a time (readWeights order)."""
    cfg = ctx.cfg
    for kind, layers, count in runtime.tensor_shapes(cfg):
        for layer in range(max(layers, 1)):
            a = orc.weights(kind, layer if layers else -1)
            assert a.size == count
            ctx.upload(kind, layer if layers else -1, a)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as graft
    graft.build()
    return runtime.lib()


@pytest.mark.parametrize("name,exact", [("tiny", 0), ("tiny", 1), ("ragged", 0), ("ragged", 1),
                                        ("stories15M", 0), ("stories15M_prompt", 0)])
def test_forward_matches_reference_goldens(built, name, exact):
    meta, g = load_gold(name)
    orc = O.Oracle(meta["header"], meta["seed"])
    ctx = runtime.Context(meta["header"])
    upload_from_oracle(ctx, orc)
    ctx.set_option(runtime.OPT_EXACT_ATTENTION, exact)
    ctx.set_option(runtime.OPT_KEEP_STATE, 1)
    keep = {p: i for i, p in enumerate(meta["logit_positions"])}
    worst, biteq, total = 0.0, 0, 0
    for pos, tok in enumerate(meta["tokens_fed"]):
        got = np.array(ctx.forward(tok, pos), copy=True)
        assert runtime.argmax(got) == meta["argmax"][pos], (name, pos)
        if pos in keep:
            want = g["logits"][keep[pos]]
            worst = max(worst, float(np.abs(got - want).max()))
            biteq += int((bits(got) == bits(want)).sum())
            total += want.size
            if "x" in g.files:   # RunState scratch buffers after the call (llama2.ts:131-146)
                for nm in ("x", "xb", "xb2", "hb", "hb2", "q", "k", "v"):
                    st = ctx.read_state(nm)
                    assert np.abs(st - g[nm][pos]).max() <= TOL, (name, pos, nm)
                att = ctx.read_state("att").reshape(ctx.cfg.n_heads, ctx.cfg.seq_len)[:, :pos + 1]
                ga = g["att"][pos].reshape(ctx.cfg.n_heads, ctx.cfg.seq_len)[:, :pos + 1]
                assert np.abs(att - ga).max() <= 1e-6, (name, pos, "att")
    assert worst <= TOL, (name, worst)
    print("\n[%s exact=%d] max|dlogit|=%.3g, bit-identical logits %.4f%% of %d" % (name, exact, worst, 100.0 * biteq / total, total))
    if exact:
        assert biteq / total > 0.999, "exact mode should reproduce the reference's roundings"
    if "key_cache" in g.files:
        n = meta["steps_run"]
        d, S, L = ctx.cfg.dim, ctx.cfg.seq_len, ctx.cfg.n_layers
        for nm in ("key_cache", "value_cache"):
            got = ctx.read_state(nm).reshape(L, S, d)[:, :n]
            want = g[nm].reshape(L, S, d)[:, :n]
            assert np.abs(got - want).max() <= TOL, nm
    ctx.close()


@pytest.mark.parametrize("name", ["stories110M", "llama2_7b_L2"])
def test_whole_context_matches_reference_goldens(built, name):
    """The TRUE reference's whole context window (1024 steps at head_size 64, 2048 at head_size 128: every
    attention split level of the HIP path and every multiple of the tile round): the argmax of every step and the
    logits at the kept positions -- first / last steps and both sides of every power of two -- through l2_forward,
    then the same token stream from the device-resident greedy loop (one captured graph per split level)."""
    meta, g = load_gold(name)
    ctx = runtime.Context(meta["header"])
    ctx.synth_fill(meta["seed"])        # device generator == oracle generator (checked below)
    keep = {p: i for i, p in enumerate(meta["logit_positions"])}
    worst = 0.0
    for pos, tok in enumerate(meta["tokens_fed"]):
        got = ctx.forward(tok, pos)
        assert runtime.argmax(got) == meta["argmax"][pos], (name, pos)
        if pos in keep:
            err = float(np.abs(got - g["logits"][keep[pos]]).max())
            assert err <= TOL, (name, pos, err)
            worst = max(worst, err)
    n = meta["steps_run"]
    assert n == ctx.cfg.seq_len and meta["tokens_fed"] == [1] + meta["argmax"][:-1]
    toks = ctx.decode_greedy(1, 0, n)
    assert toks.tolist() == meta["argmax"]
    print("\n[%s] %d steps token-exact, max|dlogit| %.3g at %d kept positions" % (name, n, worst, len(keep)))
    ctx.close()


def _random_headers(count, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < count:
        H = int(rng.choice([1, 2, 3, 4, 6, 8]))
        hs = int(rng.choice([2, 4, 6, 8, 10, 16, 22, 32, 48, 64, 128]))
        d = H * hs
        if d > 512:
            continue
        h = int(rng.integers(d // 2 + 1, 3 * d + 2))
        L = int(rng.integers(1, 4))
        V = int(rng.integers(17, 700)) * int(rng.choice([-1, 1]))
        S = int(rng.integers(3, 200))
        out.append((d, h, L, H, H, V, S))
    return out


# shapes that take a kernel instance nothing else reaches: heads wider than 128 floats (a whole wave per cache row: the 4-wave
# attention form), input widths of 1537 .. 2048 floats (w1 / w3 leave the latency form for the streaming one there, the other
# phases take the latency form's widest instance)
WIDE_SHAPES = [(512, 640, 1, 2, 2, 300, 150), (384, 1000, 2, 2, 2, -211, 40), (1792, 2304, 1, 14, 14, 257, 24), (2048, 1600, 1, 16, 16, -130, 20)]


@pytest.mark.parametrize("hdr", _random_headers(14, 20261003) + WIDE_SHAPES)
def test_random_shapes_match_the_oracle(built, hdr):
    """Shapes nobody tuned for -- odd hidden sizes, head sizes that are not powers of two or not multiples of 4 (scalar kernels),
    one head, vocabularies that are not multiples of anything, shared and unshared classifiers, contexts shorter than a tile --
    against the oracle (itself pinned to the reference): logits <= 1e-4 and the same argmax at every step of the whole context
    (up to 40 steps), l2_forward and the device greedy loop, then the same prompt through l2_prefill."""
    orc = O.Oracle(hdr, 7)
    ctx = runtime.Context(hdr)
    upload_from_oracle(ctx, orc)
    steps = min(hdr[6], 40)
    tok, fed = 1, []
    for pos in range(steps):
        fed.append(tok)
        got = ctx.forward(tok, pos)
        want = orc.forward(tok, pos)
        assert np.abs(got - want).max() <= TOL, (hdr, pos)
        assert runtime.argmax(got) == O.argmax(want), (hdr, pos)
        tok = O.argmax(want)
    assert ctx.decode_greedy(1, 0, steps).tolist() == fed[1:] + [tok]
    b = runtime.Context(hdr)
    upload_from_oracle(b, orc)
    lp = b.prefill(fed, 0)
    assert np.abs(lp - want).max() <= TOL and runtime.argmax(lp) == tok
    ctx.close(); b.close(); orc.close()


def test_full_llama2_7b_matches_reference_golden(built):
    """BASELINE.json config 4 itself -- all 32 layers, 27 GB of weights: the 256 steps bench.py times by default (round 4: 47 minutes of the
    real reference at 10 s per token, the whole file in host memory) and, where the fixture holds them (round 6: 1024 steps, three hours), on
    through every attention split level of the full model: every argmax through the drop-in call, logits at the kept positions (0, 2, 19, 63,
    127, 255 and both sides of 144, 256, 512), then the same tokens from the device-resident loop."""
    meta, g = load_gold("llama2_7b")
    ctx = runtime.Context(meta["header"])
    ctx.synth_fill(meta["seed"])
    keep = {p: i for i, p in enumerate(meta["logit_positions"])}
    for pos, tok in enumerate(meta["tokens_fed"]):
        got = ctx.forward(tok, pos)
        assert runtime.argmax(got) == meta["argmax"][pos], pos
        if pos in keep:
            assert np.abs(got - g["logits"][keep[pos]]).max() <= TOL, pos
    n = len(meta["argmax"])
    assert n >= 256 and set([0, 2, 19, 63, 127, 255]) <= set(meta["logit_positions"])
    assert ctx.decode_greedy(1, 0, n).tolist() == meta["argmax"]
    ctx.close()


def test_device_greedy_loop_is_token_exact_for_256_steps(built):
    """`-t 0 -s 1 -n 256` (package.json deterministic script, llama2.ts:465-508) kept on the device."""
    meta, _ = load_gold("stories15M")
    ctx = runtime.Context(meta["header"])
    ctx.synth_fill(meta["seed"])
    toks = ctx.decode_greedy(1, 0, 256)
    assert toks.tolist() == meta["argmax"]
    assert [1] + toks[:-1].tolist() == meta["tokens_fed"]
    ctx.close()


def test_synth_fill_equals_oracle_generator(built):
    hdr = configs.header("stories15M")
    ctx = runtime.Context(hdr)
    ctx.synth_fill(7)
    for kind, layers, count in runtime.tensor_shapes(ctx.cfg):
        for layer in ([0, layers - 1] if layers else [-1]):
            want = O.synth_tensor(hdr, 7, kind, layer)
            got = ctx.read_tensor(kind, max(layer, 0), 0, count)
            assert np.array_equal(bits(got), bits(want)), (kind, layer)
    ctx.close()


def test_upload_roundtrip_and_errors(built):
    hdr = configs.header("tiny")
    ctx = runtime.Context(hdr)
    with pytest.raises(runtime.L2Error) as e:       # forward before the weights are there
        ctx.forward(1, 0)
    assert e.value.code == -4
    a = np.arange(64 * 64, dtype=np.float32)
    ctx.upload(runtime.T_WQ, 1, a)
    assert np.array_equal(ctx.read_tensor(runtime.T_WQ, 1, 0, a.size), a)
    with pytest.raises(runtime.L2Error):
        ctx.upload(runtime.T_WQ, 1, a[:-1])          # wrong size
    with pytest.raises(runtime.L2Error):
        ctx.upload(runtime.T_WQ, 2, a)               # layer out of range
    with pytest.raises(runtime.L2Error):
        ctx.upload(runtime.T_WCLS, -1, a)            # shared classifier: no separate wcls (llama2.ts:127)
    ctx.synth_fill(1)
    with pytest.raises(runtime.L2Error):
        ctx.forward(1, 64)                           # pos == seq_len
    with pytest.raises(runtime.L2Error):
        ctx.forward(512, 0)                          # token == vocab_size
    ctx.close()


def test_graph_and_eager_launches_agree(built):
    meta, _ = load_gold("tiny")
    outs = []
    for use_graph in (1, 0):
        ctx = runtime.Context(meta["header"])
        ctx.synth_fill(meta["seed"])
        ctx.set_option(runtime.OPT_USE_GRAPH, use_graph)
        res = [np.array(ctx.forward(t, p), copy=True) for p, t in enumerate(meta["tokens_fed"][:12])]
        outs.append(np.stack(res))
        ctx.close()
    assert np.array_equal(bits(outs[0]), bits(outs[1]))


def test_long_context_full_sequence(built):
    """Maximum position (pos = seq_len-1) and size-independent properties at the full 110M shape:
    logits finite, softmax rows sum to 1, greedy continuation equals the oracle's."""
    hdr = configs.header("stories110M")
    ctx = runtime.Context(hdr)
    ctx.synth_fill(3)
    ctx.set_option(runtime.OPT_KEEP_STATE, 1)
    S = ctx.cfg.seq_len
    toks = ctx.decode_greedy(1, 0, S)
    assert toks.min() >= 0 and toks.max() < ctx.cfg.vocab_size
    att = ctx.read_state("att").reshape(ctx.cfg.n_heads, S)
    assert np.allclose(att.sum(axis=1), 1.0, atol=1e-5)     # last layer, pos = S-1: full rows
    lg = ctx.read_state("logits")
    assert np.isfinite(lg).all()
    orc = O.Oracle(hdr, 3)
    tok = 1
    for pos in range(6):
        want = orc.forward(tok, pos)
        tok = O.argmax(want)
        assert tok == toks[pos]
    ctx.close()


@pytest.mark.parametrize("name,splits", [("tiny", 4), ("ragged", 3), ("stories15M", 8)])
def test_split_attention_matches_reference(built, name, splits):
    """Flash-decode split of the timesteps over several workgroups per head (default for long contexts)."""
    meta, g = load_gold(name)
    os.environ["L2_ATTN_SPLITS"] = str(splits)
    try:
        ctx = runtime.Context(meta["header"])
    finally:
        del os.environ["L2_ATTN_SPLITS"]
    ctx.synth_fill(meta["seed"])
    ctx.set_option(runtime.OPT_KEEP_STATE, 1)
    keep = {p: i for i, p in enumerate(meta["logit_positions"])}
    for pos, tok in enumerate(meta["tokens_fed"]):
        got = np.array(ctx.forward(tok, pos), copy=True)
        assert runtime.argmax(got) == meta["argmax"][pos], (name, pos)
        if pos in keep:
            assert np.abs(got - g["logits"][keep[pos]]).max() <= TOL
            if "att" in g.files:
                H, S = ctx.cfg.n_heads, ctx.cfg.seq_len
                att = ctx.read_state("att").reshape(H, S)[:, :pos + 1]
                assert np.abs(att - g["att"][pos].reshape(H, S)[:, :pos + 1]).max() <= 1e-6
                assert np.abs(ctx.read_state("xb2") - g["xb2"][pos]).max() <= TOL
    ctx.close()


def test_long_context_crosses_every_split_level_token_exact(built):
    """1280 greedy steps at tiny width: attention runs unsplit (pos < 256) and split (thresholds scale with head size; forced 8-way below);
    the token stream and the logits at the level boundaries must still equal the TRUE reference's."""
    meta, g = load_gold("tinylong")
    ctx = runtime.Context(meta["header"])
    ctx.synth_fill(meta["seed"])
    toks = ctx.decode_greedy(1, 0, meta["steps_run"])
    assert toks.tolist() == meta["argmax"]
    ctx.close()
    ctx = runtime.Context(meta["header"])
    ctx.synth_fill(meta["seed"])
    keep = {p: i for i, p in enumerate(meta["logit_positions"])}
    for pos, tok in enumerate(meta["tokens_fed"]):
        got = ctx.forward(tok, pos)
        if pos in keep:
            assert np.abs(got - g["logits"][keep[pos]]).max() <= TOL, pos
    ctx.close()


def test_native_checkpoint_loader_equals_per_array_upload(built, tmp_path):
    """l2_load_checkpoint vs readWeights + l2_upload: same bytes in HBM, same logits."""
    import time
    meta, g = load_gold("stories15M")
    path = str(tmp_path / "m.bin")
    O.synth_write(meta["header"], meta["seed"], path)
    t0 = time.time()
    cfg, state, weights, nbytes = runtime.load_checkpoint_native(path)
    t1 = time.time()
    cfg2, state2, weights2 = runtime.load_checkpoint(path)
    t2 = time.time()
    assert nbytes == os.path.getsize(path) == 60816028 and cfg.header == tuple(meta["header"])
    for kind, layers, count in runtime.tensor_shapes(cfg):
        for layer in ([0, layers - 1] if layers else [0]):
            a = weights.ctx.read_tensor(kind, layer, 0, count)
            b = weights2.ctx.read_tensor(kind, layer, 0, count)
            assert np.array_equal(bits(a), bits(b)), (kind, layer)
    runtime.transformer(1, 0, cfg, state, weights)
    assert np.abs(state.logits - g["logits"][0]).max() <= TOL
    print("\nnative loader %.0f MB/s, readWeights+l2_upload %.0f MB/s" % (nbytes / 1e6 / (t1 - t0), nbytes / 1e6 / (t2 - t1)))
    weights.ctx.close(); weights2.ctx.close()


def test_full_size_7b_properties(built):
    """BASELINE.json's full Llama-2-7B shape (27 GB of synthetic weights generated on the device): properties that
    do not need a CPU pass over 27 GB -- the captured-graph and eager paths agree bit for bit, the bit-faithful and
    default attention agree within the gate, the split-attention launch gives the same tokens, the
    greedy stream is reproducible, and logits are finite with softmax rows summing to one."""
    hdr = configs.header("llama2_7b")
    ctx = runtime.Context(hdr)
    ctx.synth_fill(11)
    ctx.set_option(runtime.OPT_KEEP_STATE, 1)
    a = [np.array(ctx.forward(t, p), copy=True) for p, t in enumerate([1, 5, 9])]
    ctx.set_option(runtime.OPT_USE_GRAPH, 0)
    b = [np.array(ctx.forward(t, p), copy=True) for p, t in enumerate([1, 5, 9])]
    assert all(np.array_equal(bits(x), bits(y)) for x, y in zip(a, b))
    ctx.set_option(runtime.OPT_EXACT_ATTENTION, 1)
    c = [np.array(ctx.forward(t, p), copy=True) for p, t in enumerate([1, 5, 9])]
    assert max(float(np.abs(x - y).max()) for x, y in zip(a, c)) <= 1e-5
    ctx.set_option(runtime.OPT_EXACT_ATTENTION, 0)
    ctx.set_option(runtime.OPT_USE_GRAPH, 1)
    assert np.isfinite(a[2]).all()
    att = ctx.read_state("att").reshape(ctx.cfg.n_heads, ctx.cfg.seq_len)[:, :3]
    assert np.allclose(att.sum(axis=1), 1.0, atol=1e-5)
    t1 = ctx.decode_greedy(1, 0, 12)
    t2 = ctx.decode_greedy(1, 0, 12)
    assert t1.tolist() == t2.tolist()
    ctx.close()
    os.environ["L2_ATTN_SPLITS"] = "8"
    try:
        ctx = runtime.Context(hdr)
    finally:
        del os.environ["L2_ATTN_SPLITS"]
    ctx.synth_fill(11)
    t4 = ctx.decode_greedy(1, 0, 12)
    assert t4.tolist() == t1.tolist()
    ctx.close()


@pytest.mark.parametrize("name,n", [("tiny", 1), ("tiny", 5), ("tiny", 16), ("tiny", 17), ("tiny", 32), ("tiny", 33), ("tiny", 40), ("tiny", 64), ("tinylong", 65), ("tinylong", 150),
                                    ("ragged", 7), ("stories15M", 21), ("stories15M", 70)])
def test_prefill_equals_token_by_token(built, name, n):
    """l2_prefill (chunks of 16, 32 or 64 tokens on fp64 MFMA) must leave the KV cache and the last logits exactly
    where n separate transformer() calls -- what the reference does with a prompt, llama2.ts:471-473 -- leave them,
    and decoding must continue identically.  `ragged` (dims not multiples of 16) takes the fallback path."""
    meta, g = load_gold(name)
    toks = meta["tokens_fed"][:n]
    a = runtime.Context(meta["header"]); a.synth_fill(meta["seed"])
    b = runtime.Context(meta["header"]); b.synth_fill(meta["seed"])
    for pos, t in enumerate(toks):
        la = np.array(a.forward(t, pos), copy=True)
    lb = np.array(b.prefill(toks, 0), copy=True)
    assert np.abs(la - lb).max() <= 1e-5
    assert runtime.argmax(lb) == meta["argmax"][n - 1]
    keep = {p: i for i, p in enumerate(meta["logit_positions"])}
    if n - 1 in keep:
        assert np.abs(lb - g["logits"][keep[n - 1]]).max() <= TOL        # the TRUE reference's logits
    d, S, L = b.cfg.dim, b.cfg.seq_len, b.cfg.n_layers
    for nm in ("key_cache", "value_cache"):
        ca = a.read_state(nm).reshape(L, S, d)[:, :n]
        cb = b.read_state(nm).reshape(L, S, d)[:, :n]
        assert np.abs(ca - cb).max() <= 1e-6, nm
    ta, tb = runtime.argmax(la), runtime.argmax(lb)
    for pos in range(n, min(n + 6, S)):                                   # keep decoding on both
        la = np.array(a.forward(ta, pos), copy=True)
        lb = np.array(b.forward(tb, pos), copy=True)
        assert np.abs(la - lb).max() <= 1e-5
        ta, tb = runtime.argmax(la), runtime.argmax(lb)
        assert ta == tb == meta["argmax"][pos]
    a.close(); b.close()


@pytest.mark.parametrize("form", ["register-blocked", "16-row tiles"])
@pytest.mark.parametrize("name,n", [("stories110M", 128), ("stories110M", 256), ("llama2_7b_L2", 64), ("llama2_7b_L2", 128), ("llama2_7b_L2", 256)])
def test_prefill_at_the_widths_the_bench_times(built, name, n, form, monkeypatch):
    """Prompt ingestion where bench.py reports `prefill_tok_s` (d = 4096 / h = 11008, and the 110M width): the first n
    tokens the TRUE reference fed itself (llama2.ts:471-473 teacher-forces a prompt one transformer() call per token) go
    through l2_prefill in one call; the logits of position n - 1 must be the reference's own (kept positions 63 / 127 /
    255, <= 1e-4), the KV cache must equal the token-by-token path's, and the greedy continuation must follow the
    reference's tokens through the next kept position.  Every form of the GEMMs: the register-blocked kernels (default: up to
    four 64-token chunks per launch) and the older 16-row-tile kernels with and without the LDS weight tile."""
    monkeypatch.setenv("L2_PF3", "1" if form == "register-blocked" else "0")
    meta, g = load_gold(name)
    keep = {p: i for i, p in enumerate(meta["logit_positions"])}
    assert n - 1 in keep, "fixture keeps no logits at %d" % (n - 1)
    toks = meta["tokens_fed"][:n]
    b = runtime.Context(meta["header"]); b.synth_fill(meta["seed"])
    lb = np.array(b.prefill(toks, 0), copy=True)
    err = float(np.abs(lb - g["logits"][keep[n - 1]]).max())
    assert err <= TOL, (name, n, err)
    assert runtime.argmax(lb) == meta["argmax"][n - 1]
    a = runtime.Context(meta["header"]); a.synth_fill(meta["seed"])
    for pos, t in enumerate(toks):
        a.forward(t, pos)
    d, S, L = b.cfg.dim, b.cfg.seq_len, b.cfg.n_layers
    for nm in ("key_cache", "value_cache"):
        ca = a.read_state(nm).reshape(L, S, d)[:, :n]
        cb = b.read_state(nm).reshape(L, S, d)[:, :n]
        assert np.abs(ca - cb).max() <= 1e-6, nm
    a.close()
    nxt = min(p for p in keep if p >= n)                      # the next kept position: n itself (128 -> 128, 256 -> 256) or 64 -> 64
    tok = runtime.argmax(lb)
    for pos in range(n, nxt + 1):
        assert tok == meta["tokens_fed"][pos], (name, pos)
        lg = b.forward(tok, pos)
        tok = runtime.argmax(lg)
        assert tok == meta["argmax"][pos]
    assert np.abs(lg - g["logits"][keep[nxt]]).max() <= TOL
    # and on from there with the device loop: token-exact for 32 more positions
    cont = b.decode_greedy(tok, nxt + 1, 32)
    assert cont.tolist() == meta["argmax"][nxt + 1:nxt + 33]
    print("\n[prefill %s n=%d %s] max|dlogit| vs reference %.3g" % (name, n, form, err))
    b.close()


@pytest.mark.parametrize("name,n", [("stories110M", 128), ("stories110M", 256), ("llama2_7b_L2", 64), ("llama2_7b_L2", 128), ("llama2_7b_L2", 256)])
def test_prefill_fp32_accumulate_is_an_opt_in_within_the_logit_bar(built, name, n):
    """L2_OPT_PREFILL_F32_MFMA (build both, choose by evidence: SURVEY.md section 7): the register-blocked prompt GEMMs on
    v_mfma_f32_16x16x4_f32 -- fp32 accumulate, NOT the reference's arithmetic (llama2.ts:196-203 accumulates in a double), so NOT the
    default.  What it must still do: the reference's logits within 1e-4 at the kept positions, the reference's argmax there, a KV
    cache within fp32 accumulation error of the default form's -- and leave the default alone (option off: bit for bit the fp64 form)."""
    meta, g = load_gold(name)
    keep = {p: i for i, p in enumerate(meta["logit_positions"])}
    toks = meta["tokens_fed"][:n]
    a = runtime.Context(meta["header"]); a.synth_fill(meta["seed"])
    assert a.get_option(runtime.OPT_PREFILL_F32_MFMA) == 0
    la = np.array(a.prefill(toks, 0), copy=True)
    b = runtime.Context(meta["header"]); b.synth_fill(meta["seed"])
    b.set_option(runtime.OPT_PREFILL_F32_MFMA, 1)
    lb = np.array(b.prefill(toks, 0), copy=True)
    err = float(np.abs(lb - g["logits"][keep[n - 1]]).max())
    assert err <= TOL and runtime.argmax(lb) == meta["argmax"][n - 1], (name, n, err)
    assert not np.array_equal(bits(la), bits(lb)), "the fp32 form produced the fp64 form's bits: the option did nothing"
    d, S, L = b.cfg.dim, b.cfg.seq_len, b.cfg.n_layers
    for nm in ("key_cache", "value_cache"):
        ca = a.read_state(nm).reshape(L, S, d)[:, :n]
        cb = b.read_state(nm).reshape(L, S, d)[:, :n]
        assert np.abs(ca - cb).max() <= 2e-5, nm
    b.set_option(runtime.OPT_PREFILL_F32_MFMA, 0)
    assert np.array_equal(bits(np.array(b.prefill(toks, 0), copy=True)), bits(la))
    print("\n[prefill fp32 accumulate %s n=%d] max|dlogit| vs reference %.3g (fp64 form %.3g)" % (name, n, err, float(np.abs(la - g["logits"][keep[n - 1]]).max())))
    a.close(); b.close()


@pytest.mark.parametrize("n,first", [(65, 0), (70, 0), (129, 0), (200, 0), (160, 100), (300, 37)])
def test_prefill_ragged_chunk_counts_at_110m_width(built, n, first):
    """The register-blocked GEMMs take up to four 64-token chunks per launch: prompts that end inside a chunk (65, 70, 129,
    200 tokens), that are fed in two calls (the second starting at pos0 = `first`: llama2.ts:471-473 has no notion of calls, only
    of positions) or that exceed one launch sequence (300) must leave the same KV cache and last logits as one transformer()
    call per token, and the reference's own argmax where its fixture has one."""
    meta, g = load_gold("stories110M")
    toks = meta["tokens_fed"][:n]
    a = runtime.Context(meta["header"]); a.synth_fill(meta["seed"])
    b = runtime.Context(meta["header"]); b.synth_fill(meta["seed"])
    for pos, t in enumerate(toks):
        la = np.array(a.forward(t, pos), copy=True)
    if first:
        b.prefill(toks[:first], 0)
    lb = np.array(b.prefill(toks[first:], first), copy=True)
    assert np.abs(la - lb).max() <= 1e-5
    assert runtime.argmax(lb) == meta["argmax"][n - 1]
    d, S, L = b.cfg.dim, b.cfg.seq_len, b.cfg.n_layers
    for nm in ("key_cache", "value_cache"):
        ca = a.read_state(nm).reshape(L, S, d)[:, :n]
        cb = b.read_state(nm).reshape(L, S, d)[:, :n]
        assert np.abs(ca - cb).max() <= 1e-6, nm
    assert b.decode_greedy(runtime.argmax(lb), n, 8).tolist() == meta["argmax"][n:n + 8]
    a.close(); b.close()


def test_prefill_of_the_whole_7b_width_context(built):
    """All 2048 positions of the 7B-width golden through l2_prefill in ONE call (eight launch sequences of 256; the last query
    tile sees 2048 keys: 131 KB of score rows in LDS): logits of the last position and of position 1919 (a second context fed
    1920 tokens) against the TRUE reference, the KV cache of the last 128 positions against token-by-token decoding from 1920 on."""
    meta, g = load_gold("llama2_7b_L2")
    keep = {p: i for i, p in enumerate(meta["logit_positions"])}
    fed = meta["tokens_fed"]
    a = runtime.Context(meta["header"]); a.synth_fill(meta["seed"])
    la = np.array(a.prefill(fed[:2048], 0), copy=True)
    assert np.abs(la - g["logits"][keep[2047]]).max() <= TOL and runtime.argmax(la) == meta["argmax"][2047]
    b = runtime.Context(meta["header"]); b.synth_fill(meta["seed"])
    lb = np.array(b.prefill(fed[:1920], 0), copy=True)
    assert np.abs(lb - g["logits"][keep[1919]]).max() <= TOL
    tok = runtime.argmax(lb)
    for pos in range(1920, 2048):                     # the rest one transformer() call per token, as the reference does
        assert tok == fed[pos]
        lb = b.forward(tok, pos)
        tok = runtime.argmax(lb)
    assert np.abs(lb - la).max() <= 1e-5
    d, S, L = a.cfg.dim, a.cfg.seq_len, a.cfg.n_layers
    for nm in ("key_cache", "value_cache"):
        ca = a.read_state(nm).reshape(L, S, d)[:, 1920:]
        cb = b.read_state(nm).reshape(L, S, d)[:, 1920:]
        assert np.abs(ca - cb).max() <= 1e-6, nm
    a.close(); b.close()


def test_prefill_prompt_golden_and_errors(built):
    meta, g = load_gold("stories15M_prompt")          # -i "Once upon a time": BOS + 4 prompt ids are teacher-forced
    ctx = runtime.Context(meta["header"]); ctx.synth_fill(meta["seed"])
    lg = np.array(ctx.prefill(meta["tokens_fed"][:5], 0), copy=True)
    keep = {p: i for i, p in enumerate(meta["logit_positions"])}
    assert np.abs(lg - g["logits"][keep[4]]).max() <= TOL
    tok = runtime.argmax(lg)
    for pos in range(5, 24):
        assert tok == meta["tokens_fed"][pos]
        lg = ctx.forward(tok, pos)
        tok = runtime.argmax(lg)
    assert np.abs(lg - g["logits"][keep[23]]).max() <= TOL
    with pytest.raises(runtime.L2Error):
        ctx.prefill([1, 2, 3], 255)                    # runs past seq_len
    with pytest.raises(runtime.L2Error):
        ctx.prefill([1, 40000], 0)                     # token out of range
    ctx.close()


# ---- SURVEY.md 8(f1), the rest of it: temperature / top-p sampling on the device -------------------------------

def _sampled_run(name):
    meta = json.load(open(os.path.join(GOLD, name + ".json")))
    argv = dict(zip(meta["argv"][::2], meta["argv"][1::2]))
    return meta, float(argv.get("-t", 1.0)), float(argv.get("-p", 1.0)), int(argv["-s"])


@pytest.mark.parametrize("name,n_prompt", [("cli_temp", 0), ("cli_topp", 4)])
def test_device_sampler_reproduces_the_reference_run(name, n_prompt):
    """The TRUE reference was run with -t / -p / -s as recorded in the fixture.  Forcing its prompt tokens and then
    letting l2_decode_sample pick every following token (device softmax, sequential running sums, stable sort,
    xorshift* draws) must give the reference's token ids, in one call and in ragged chunks (RNG state carried)."""
    meta, temperature, topp, seed = _sampled_run(name)
    fed = meta["tokens_fed"]
    for chunks in ([len(fed) - 1 - n_prompt], [1, 2, 5, 10 ** 6]):
        ctx = runtime.Context(meta["header"])
        ctx.synth_fill(meta["seed"])
        for pos in range(n_prompt):
            ctx.forward(fed[pos], pos)                       # teacher-forced prompt positions (llama2.ts:471-473)
        pos, tok, rng, got = n_prompt, fed[n_prompt], seed, []
        for n in chunks:
            n = min(n, len(fed) - 1 - pos)
            if n <= 0:
                break
            toks, rng = ctx.decode_sample(tok, pos, n, temperature, topp, rng)
            got += toks.tolist()
            pos += n
            tok = got[-1]
        assert got == fed[n_prompt + 1:], (name, chunks)
        ctx.close()


def test_device_sampler_matches_oracle_on_other_settings():
    """Settings the fixtures do not cover, checked against the oracle's sampler fed with the device's own logits:
    low / high temperature, top-p that keeps a handful or nearly all of the vocabulary, topp outside (0, 1)."""
    hdr = configs.header("stories15M")
    for temperature, topp, seed in [(0.3, 1.0, 5), (1.7, 0.0, 99), (1.0, 0.05, 3), (0.8, 0.999, 123456789), (2.5, 0.5, 2 ** 63 + 11)]:
        ctx = runtime.Context(hdr)
        ctx.synth_fill(1)
        toks, rng_after = ctx.decode_sample(1, 0, 12, temperature, topp, seed)
        ref = runtime.Context(hdr)
        ref.synth_fill(1)
        rng = O.Rng(seed)
        tok, want = 1, []
        for pos in range(12):
            lg = np.array(ref.forward(tok, pos), copy=True)
            tok, _ = O.next_token(lg, temperature, topp, rng)
            want.append(tok)
        assert toks.tolist() == want, (temperature, topp, seed)
        assert rng_after == rng.state.value
        ctx.close(); ref.close()


@pytest.mark.parametrize("hdr,why", [((64, 176, 1, 4, 4, 50257, 16), "50 sorted tiles: seven groups of the rank merge, the last one ragged"),
                                     ((64, 176, 1, 4, 4, 128256, 16), "126 sorted tiles in 16 groups (a Llama-3 vocabulary)"),
                                     ((64, 176, 1, 4, 4, 38912, 16), "38 sorted tiles"),
                                     ((64, 176, 1, 4, 4, 1000, 16), "one ragged tile"),
                                     ((64, 176, 1, 4, 4, 5121, 16), "last tile holds one element")])
def test_device_sampler_other_vocabularies(hdr, why):
    """Vocabulary sizes around the sampler's tile and LDS limits, plain and top-p, negative temperature (the reference
    divides by whatever it is given, llama2.ts:482; the maximum then comes from the sampler's own pass instead of the
    classifier's argmax keys): tokens and RNG state equal the oracle's sampler fed with the device's own logits."""
    for temperature, topp, seed in [(0.9, 1.0, 11), (1.1, 0.9, 12), (-0.8, 0.95, 13), (0.6, 0.3, 14)]:
        ctx = runtime.Context(hdr)
        ctx.synth_fill(3)
        toks, rng_after = ctx.decode_sample(1, 0, 8, temperature, topp, seed)
        ref = runtime.Context(hdr)
        ref.synth_fill(3)
        rng = O.Rng(seed)
        tok, want = 1, []
        for pos in range(8):
            lg = np.array(ref.forward(tok, pos), copy=True)
            tok, _ = O.next_token(lg, temperature, topp, rng)
            want.append(tok)
        assert toks.tolist() == want, (why, temperature, topp)
        assert rng_after == rng.state.value
        ctx.close(); ref.close()


def test_device_sampler_long_runs_against_oracle():
    """Hundreds of sampled tokens per setting (plain and top-p, several temperatures) against the oracle's sampler fed with the
    device's own logits: every token's running sums go through the tiled runs / chain / search with whatever ties, binade
    crossings and nucleus sizes those distributions produce; one flipped index anywhere changes every later token.  The two huge temperatures flatten the distribution until thousands of
    probabilities are EQUAL in fp32: the order inside such a tie (by token id, the stable sort of llama2.ts:380) decides the token."""
    hdr = configs.header("stories15M")
    for temperature, topp, seed, n in [(1.0, 0.95, 2024, 200), (0.7, 1.0, 7, 200), (1.5, 0.6, 99, 120), (0.05, 0.9, 5, 60),
                                      (1e6, 0.5, 11, 40), (3e7, 0.999, 12, 40)]:
        ctx = runtime.Context(hdr)
        ctx.synth_fill(2)
        toks, rng_after = ctx.decode_sample(1, 0, n, temperature, topp, seed)
        ref = runtime.Context(hdr)
        ref.synth_fill(2)
        rng = O.Rng(seed)
        tok, want = 1, []
        for pos in range(n):
            lg = np.array(ref.forward(tok, pos), copy=True)
            tok, _ = O.next_token(lg, temperature, topp, rng)
            want.append(tok)
        assert toks.tolist() == want, (temperature, topp, seed, int(np.argmax(np.array(toks.tolist()) != np.array(want))))
        assert rng_after == rng.state.value
        # picked by the margin rule, not by its serial loop (a few tokens in a million would be; equal probabilities by the thousand -- the
        # two huge temperatures -- put many running sums near a threshold at once and are allowed more)
        assert ctx.get_option(runtime.OPT_SAMPLED_TOKENS) == n and ctx.get_option(runtime.OPT_SAMPLED_SERIAL) <= (n // 4 if temperature >= 1e6 else 1)
        ctx.close(); ref.close()


def test_device_sampler_temperature_zero_is_greedy():
    ctx = runtime.Context(configs.header("tiny"))
    ctx.synth_fill(1)
    a, rng = ctx.decode_sample(1, 0, 16, 0.0, 0.9, 77)
    ctx2 = runtime.Context(configs.header("tiny"))
    ctx2.synth_fill(1)
    assert a.tolist() == ctx2.decode_greedy(1, 0, 16).tolist() and rng == 77
    ctx.close(); ctx2.close()


_serial_sums = sum_cases.serial_sums


def test_exact_parallel_running_sums_equal_the_serial_loop():
    """The sampler's core claim: its parallel accumulation is BIT-identical to the serial fp64 loop.  Vectors built to
    hit every branch: softmax-like tails, exact ties at every distance below the grid, sums crossing powers of two
    (also exactly onto one), zeros and subnormals in front, one huge element late, constant runs, windows > 4096."""
    cases = sum_cases.adversarial()
    for n, v in enumerate(cases):
        got = runtime.running_sums(v)
        want = _serial_sums(v)
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), (n, int(np.argmax(got != want)))


def test_sampler_forms_agree(monkeypatch):
    """The device sampler's forms give the same tokens and RNG state: the default (tree sums + the proven margin, csrc/sampler_margin.hip.h),
    the same with every token declared undecided (the reference's loop run as written by one lane: the branch a token takes when a running sum
    comes within the margin of its threshold), every running sum exact on the whole chip (the round-2/3 default) and the one-workgroup serial
    form.  The counters say which branch picked: none by the serial loop in the default form here, all of them when forced."""
    hdr = configs.header("stories15M")
    settings = ((0.9, 1.0), (1.3, 0.8), (1e6, 0.5), (0.2, 0.999), (-0.8, 1.0))
    runs = {}
    for form, env in (("margin", {}), ("forced", {"L2_SAMPLER_FORCE_SERIAL": "1"}), ("chain", {"L2_SAMPLER_CHAIN": "1"}), ("serial", {"L2_SAMPLER_SERIAL": "1"})):
        for k in ("L2_SAMPLER_FORCE_SERIAL", "L2_SAMPLER_CHAIN", "L2_SAMPLER_SERIAL"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctx = runtime.Context(hdr)
        ctx.synth_fill(1)
        runs[form] = [tuple(x.tolist() if hasattr(x, "tolist") else x for x in ctx.decode_sample(1, 0, 24, t, p, 9)) for t, p in settings]
        picked, by_loop = ctx.get_option(runtime.OPT_SAMPLED_TOKENS), ctx.get_option(runtime.OPT_SAMPLED_SERIAL)
        if form == "margin":
            assert picked == 24 * len(settings) and by_loop == 0
        if form == "forced":
            assert picked == 24 * len(settings) and by_loop == picked
        ctx.close()
    assert runs["margin"] == runs["forced"] == runs["chain"] == runs["serial"]


@pytest.mark.parametrize("vocab", [1000, 5121, 50257])
def test_margin_sampler_serial_branch_on_ragged_vocabularies(monkeypatch, vocab):
    """The margin form's serial loop (the branch an undecided token takes) on vocabularies that end inside a tile, inside one of its 4096-value
    staging chunks and inside a 512-value segment of recorded sums: same tokens and RNG state as the default branch, plain and top-p."""
    hdr = (64, 176, 1, 4, 4, vocab, 16)
    runs = {}
    for forced in ("0", "1"):
        monkeypatch.setenv("L2_SAMPLER_FORCE_SERIAL", forced)
        ctx = runtime.Context(hdr)
        ctx.synth_fill(3)
        runs[forced] = [tuple(x.tolist() if hasattr(x, "tolist") else x for x in ctx.decode_sample(1, 0, 12, t, p, 21)) for t, p in ((0.9, 1.0), (1.1, 0.9), (0.6, 0.3))]
        assert ctx.get_option(runtime.OPT_SAMPLED_SERIAL) == (36 if forced == "1" else 0)
        ctx.close()
    assert runs["0"] == runs["1"]


@pytest.mark.parametrize("form", ["L2_SAMPLER_FORCE_SERIAL", "L2_SAMPLER_CHAIN"])
def test_device_sampler_other_forms_reproduce_the_reference_run(monkeypatch, form):
    """The reference's own -t / -p runs (fixtures cli_temp, cli_topp) through the margin form's serial branch and through the exact chain."""
    monkeypatch.setenv(form, "1")
    for name, n_prompt in (("cli_temp", 0), ("cli_topp", 4)):
        meta, temperature, topp, seed = _sampled_run(name)
        fed = meta["tokens_fed"]
        ctx = runtime.Context(meta["header"])
        ctx.synth_fill(meta["seed"])
        for pos in range(n_prompt):
            ctx.forward(fed[pos], pos)
        toks, _ = ctx.decode_sample(fed[n_prompt], n_prompt, len(fed) - 1 - n_prompt, temperature, topp, seed)
        assert toks.tolist() == fed[n_prompt + 1:], (form, name)
        ctx.close()


@pytest.mark.parametrize("name,env", [("stories110M", {"L2_SMALL_MAX": "0"}), ("stories15M", {"L2_SMALL_MAX": "0"}),
                                      ("llama2_7b_L2", {"L2_TUNE_ROT": "0"}), ("llama2_7b_L2", {"L2_TUNE_ROT": "3"}),
                                      ("llama2_7b_L2", {"L2_PACKED": "0"}), ("llama2_7b_L2", {"L2_PACKED": "0", "L2_TUNE_ROT": "0"}),
                                      ("stories110M", {"L2_SMALL_MAX": "0", "L2_TUNE_ROT": "7"})])
def test_launch_geometry_variants_match_reference(monkeypatch, name, env):
    """The streaming form of the GEMV phases on shapes that default to the latency form, other starting columns of its rows, and the
    row-major tensors streamed instead of their repacked copies (the default at this width): same goldens, same tolerance,
    tokens exact."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    meta, g = load_gold(name)
    ctx = runtime.Context(meta["header"])
    ctx.synth_fill(meta["seed"])
    kept = meta.get("logit_positions") or list(range(len(g["logits"])))
    for pos, tok in enumerate(meta["tokens_fed"]):
        lg = ctx.forward(tok, pos)
        assert runtime.argmax(lg) == meta["argmax"][pos]
        if pos in kept:
            assert np.abs(lg - g["logits"][kept.index(pos)]).max() <= 1e-4
    n = len(meta["tokens_fed"])
    if meta["tokens_fed"] == [1] + meta["argmax"][:n - 1]:
        assert ctx.decode_greedy(1, 0, n).tolist() == meta["argmax"][:n]       # graph replay path too
    ctx.close()


def test_one_copy_of_the_weights(built, monkeypatch):
    """After the first step a matrix the streaming kernels read exists ONCE on the device: its row-major tensor is given back when the
    repacked copy is built (7B width: 2 layers + classifier).  The prompt GEMMs read the repacked copy (every kernel form: a short
    chunk through the 16-row-tile kernel, 64-token chunks through the register-blocked one), l2_read_tensor and a later l2_upload
    get the row-major bytes back out of it, and the step after that gives them away again."""
    meta, g = load_gold("llama2_7b_L2")
    hdr = meta["header"]
    ckpt_mib = configs.checkpoint_bytes(tuple(hdr)) >> 20
    ctx = runtime.Context(hdr)
    ctx.synth_fill(meta["seed"])
    assert abs(ctx.get_option(runtime.OPT_WEIGHT_MIB) - ckpt_mib) <= 2 and ctx.get_option(runtime.OPT_PACKED_MIB) == 0
    w2_before = ctx.read_tensor(runtime.T_W2, 1, 4096 * 11008 - 5000, 5000)
    wq_before = ctx.read_tensor(runtime.T_WQ, 0, 12345, 4096)
    assert runtime.argmax(ctx.forward(1, 0)) == meta["argmax"][0]
    packed = ctx.get_option(runtime.OPT_PACKED_MIB)
    total = ctx.get_option(runtime.OPT_WEIGHT_MIB)
    emb_mib = (32000 * 4096 * 4) >> 20
    assert packed >= ckpt_mib - emb_mib - 4 and abs(total - ckpt_mib) <= 4, (packed, total, ckpt_mib)      # everything but the embedding table is repacked, nothing twice
    # prompt ingestion from the repacked copy: the reference's first 70 tokens (64 + 6: register-blocked chunk, then a 16-row-tile one)
    fed = meta["tokens_fed"]
    b = runtime.Context(hdr); b.synth_fill(meta["seed"])
    lp = b.prefill(fed[:64], 0)
    assert runtime.argmax(lp) == meta["argmax"][63] and np.abs(lp - g["logits"][meta["logit_positions"].index(63)]).max() <= TOL
    lp = b.prefill(fed[64:70], 64)
    assert runtime.argmax(lp) == meta["argmax"][69]
    assert b.decode_greedy(meta["argmax"][69], 70, 20).tolist() == meta["argmax"][70:90]
    assert abs(b.get_option(runtime.OPT_WEIGHT_MIB) - ckpt_mib) <= 4
    b.close()
    # the row-major bytes come back out of the repacked copy ...
    assert np.array_equal(ctx.read_tensor(runtime.T_W2, 1, 4096 * 11008 - 5000, 5000), w2_before)
    assert np.array_equal(ctx.read_tensor(runtime.T_WQ, 0, 12345, 4096), wq_before)
    assert ctx.get_option(runtime.OPT_WEIGHT_MIB) > ckpt_mib + packed - emb_mib - 8           # both copies for the moment
    ctx.forward(1, 0)                                                                         # ... the next step gives them away again, upload or not
    assert abs(ctx.get_option(runtime.OPT_WEIGHT_MIB) - ckpt_mib) <= 4
    # ... a matrix uploaded now replaces its slice of the repacked copy at the next step, which gives the row-major tensors away again
    wo1 = ctx.read_tensor(runtime.T_WO, 1, 0, 4096 * 4096)
    ctx.upload(runtime.T_WO, 1, np.zeros(4096 * 4096, dtype=np.float32))
    z = ctx.forward(1, 0)
    assert abs(ctx.get_option(runtime.OPT_WEIGHT_MIB) - ckpt_mib) <= 4
    ctx.upload(runtime.T_WO, 1, wo1)
    again = ctx.forward(1, 0)
    assert not np.array_equal(z, again) and np.abs(again - g["logits"][0]).max() <= TOL
    ctx.close()
    # the A/B switch keeps both copies
    monkeypatch.setenv("L2_ONE_COPY", "0")
    c2 = runtime.Context(hdr); c2.synth_fill(meta["seed"]); c2.forward(1, 0)
    assert c2.get_option(runtime.OPT_WEIGHT_MIB) > ckpt_mib + packed - 8
    c2.close()


FUSED_SHAPES = [(288, 768, 2, 6, 6, 331, 300), (768, 2048, 2, 12, 12, -259, 290), (512, 1000, 3, 8, 8, 400, 70), (960, 1536, 1, 20, 20, -300, 40),
                (320, 700, 2, 5, 5, 257, 24)]


@pytest.mark.parametrize("hdr", FUSED_SHAPES)
@pytest.mark.parametrize("env", [{}, {"L2_FUSE_MIN_ROWS": "0"}, {"L2_FUSE_MIN_ROWS": "0", "L2_FUSE_FOUR_WAVES": "0"},
                                 {"L2_FUSE_SPLITS": "1", "L2_ATTN_SPLIT_ROWS": "20"}, {"L2_FUSE_SPLITS": "1", "L2_ATTN_SPLITS": "3"}])
def test_fused_qkv_attention_launch_equals_the_two_launches(built, hdr, env, monkeypatch):
    """The head-local edge inside ONE launch (attention.hip.h: qkv_attn_small_kernel; llama2.ts:216-240 -> 244-267): q, k, v of the
    position handed to the attention workgroups of the same launch as tagged granules, row pos scored from them.  Shapes that take
    it (input vectors of 257 .. 1024 floats, heads of 33 .. 64) against a context that runs the two launches (L2_FUSE_QKV_ATTN=0) and
    against the oracle: every RunState field the two phases write (q, k, v, att, the attention output xb through the wo result, the
    cache rows), logits, argmax, the device loop (one hipGraph per token: the launch counters the tags come from live on the
    device), a second run from position 0 over the same context (tags must not repeat), the unsplit level up to its 256 rows
    (four attention waves up to 128 rows, eight beyond; the fused launch from the first row on where the default starts it at 129)
    and -- behind its switch -- the fused form with a head split over several workgroups."""
    orc = O.Oracle(hdr, 5)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    fused = runtime.Context(hdr)
    upload_from_oracle(fused, orc)
    monkeypatch.setenv("L2_FUSE_QKV_ATTN", "0")
    plain = runtime.Context(hdr)
    upload_from_oracle(plain, orc)
    monkeypatch.delenv("L2_FUSE_QKV_ATTN")
    for c in (fused, plain):
        c.set_option(runtime.OPT_KEEP_STATE, 1)
    steps = hdr[6]
    tok, fed = 1, []
    for pos in range(steps):
        fed.append(tok)
        a, b = fused.forward(tok, pos), plain.forward(tok, pos)
        check = pos < 40 or pos % 37 == 0 or pos == steps - 1
        if check:
            want = orc.forward(tok, pos)
            assert np.abs(a - want).max() <= TOL and runtime.argmax(a) == O.argmax(want), (hdr, pos)
        else:
            orc.forward(tok, pos)
        assert np.abs(a - b).max() <= 2e-6, (hdr, pos)
        if check:
            for f in ("q", "k", "v", "xb2"):
                assert np.array_equal(fused.read_state(f), plain.read_state(f)) or np.abs(fused.read_state(f) - plain.read_state(f)).max() <= 1e-6, (f, pos)
            H, S = hdr[3], hdr[6]
            fa, pa = fused.read_state("att").reshape(H, S)[:, :pos + 1], plain.read_state("att").reshape(H, S)[:, :pos + 1]
            assert np.abs(fa - pa).max() <= 1e-6, pos
        tok = runtime.argmax(b)
    for f in ("key_cache", "value_cache"):      # layer 0 sees the same x in both forms: bit for bit; deeper layers inherit the attention output's last-bit differences
        assert np.array_equal(fused.read_state(f, 0), plain.read_state(f, 0)), f
        assert np.abs(fused.read_state(f) - plain.read_state(f)).max() <= 2e-6, f
    want_toks = fed[1:] + [tok]
    assert fused.decode_greedy(1, 0, steps).tolist() == want_toks          # graph replay
    assert fused.decode_greedy(1, 0, min(steps, 50)).tolist() == want_toks[:min(steps, 50)]     # again from position 0: fresh tags
    lg = fused.forward(fed[3], 3)                                         # a position fed twice in a row with different tokens
    lg2 = fused.forward(5, 3)
    pb = plain.forward(5, 3)
    assert np.abs(lg2 - pb).max() <= 2e-6 and not np.array_equal(lg, lg2)
    fused.close(); plain.close(); orc.close()


@pytest.mark.parametrize("hdr", [(1280, 2560, 2, 10, 10, -1000, 48), (1280, 2572, 2, 10, 10, 1000, 48), (2048, 5632, 1, 16, 16, -777, 32),
                                 (1280, 1280, 1, 10, 10, -140001, 16)])
def test_repacked_matrices_equal_the_row_major_ones_and_follow_uploads(built, hdr, monkeypatch):
    """The streaming form reads a second copy of its matrices, repacked on the device in the order the chip consumes them
    (kernels.hip.h, pack_kernel): every column batch, short last batches (1280 = 2.5 batches of 512 columns), short last rounds
    of row groups (2572 rows on 1288 waves), a phase that cannot be packed beside ones that are (2572 columns), a classifier
    with more than 65535 row groups and an odd row count, against the
    oracle and BIT for bit against a context that streams the row-major tensors -- then one matrix of every phase is uploaded
    again and the packed copy has to follow."""
    monkeypatch.setenv("L2_SMALL_MAX", "0")          # the streaming form at widths that default to the latency form
    orc = O.Oracle(hdr, 11)
    a = runtime.Context(hdr)
    upload_from_oracle(a, orc)
    monkeypatch.setenv("L2_PACKED", "0")
    b = runtime.Context(hdr)
    upload_from_oracle(b, orc)
    monkeypatch.delenv("L2_PACKED")

    def run(first_pos, steps, check_oracle):
        tok = 1
        for pos in range(first_pos, first_pos + steps):
            la, lb = a.forward(tok, pos), b.forward(tok, pos)
            assert np.array_equal(bits(la), bits(lb)), (hdr, pos)
            if check_oracle:
                want = orc.forward(tok, pos)
                assert np.abs(la - want).max() <= TOL and runtime.argmax(la) == O.argmax(want), (hdr, pos)
            tok = runtime.argmax(la)
        return la
    assert a.get_option(runtime.OPT_PACKED_MIB) == 0          # built with the first step
    before = run(0, 6, True)
    assert a.get_option(runtime.OPT_PACKED_MIB) > 0 and b.get_option(runtime.OPT_PACKED_MIB) == 0
    with pytest.raises(runtime.L2Error):
        a.set_option(runtime.OPT_PACKED_MIB, 1)
    assert a.decode_greedy(1, 0, 12).tolist() == b.decode_greedy(1, 0, 12).tolist()
    rng = np.random.default_rng(5)
    cfg = a.cfg
    for kind in (runtime.T_WQ, runtime.T_WO, runtime.T_W3, runtime.T_W2, runtime.T_WCLS if hdr[5] < 0 else runtime.T_TOKEN_EMBEDDING):
        layers = dict((k, l) for k, l, _ in runtime.tensor_shapes(cfg))[kind]
        layer = 1 if layers > 1 else (0 if layers else -1)
        w = orc.weights(kind, layer).copy()
        w *= rng.uniform(0.5, 1.5, size=w.shape).astype(np.float32)
        a.upload(kind, layer, w); b.upload(kind, layer, w)
    after = run(0, 6, False)
    assert not np.array_equal(bits(before), bits(after))
    if hdr[5] > 0:
        # a shared classifier's matrix is the embedding table, which is never given away: uploading it again AFTER a step dirties the
        # classifier phase only, while the layer matrices exist as repacked copies alone (round 4 refused the next step) ...
        emb = orc.weights(runtime.T_TOKEN_EMBEDDING, -1).copy() * np.float32(0.5)
        a.upload(runtime.T_TOKEN_EMBEDDING, -1, emb); b.upload(runtime.T_TOKEN_EMBEDDING, -1, emb)
        halved = run(0, 3, False)
        assert not np.array_equal(bits(halved), bits(after))
        # ... and a released matrix uploaded while the classifier phase is still stale comes back out of its own (valid) repacked copy
        emb = orc.weights(runtime.T_TOKEN_EMBEDDING, -1).copy() * np.float32(0.75)
        a.upload(runtime.T_TOKEN_EMBEDDING, -1, emb); b.upload(runtime.T_TOKEN_EMBEDDING, -1, emb)
        wq = orc.weights(runtime.T_WQ, 0).copy() * np.float32(1.25)
        a.upload(runtime.T_WQ, 0, wq); b.upload(runtime.T_WQ, 0, wq)
        run(0, 3, False)
    a.close(); b.close(); orc.close()
