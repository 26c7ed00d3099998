"""The decode step on the library's own AQL queue (csrc/aql_queue.h; llama2.ts:465-508 for the greedy loop, :468 for the blocking
call): a token's launches as hand-written packets with an agent-scope release and NO acquire fence between them, the kernels
loading every byte an earlier launch wrote past L1 (kernels.hip.h: the coherence rule).  Everything it decodes must equal what the
same context decodes through replayed hipGraphs (the runtime's own fences around every kernel node) and the real reference's goldens."""
import json
import os

import numpy as np
import pytest

from llama2_ts_amd import configs, runtime

# under a profiler's tool library the library stands down to replayed hipGraphs (the tool wraps every HSA queue and rocprofv3 crashes on
# hand-written packets): nothing to test here then
_TOOLS = [k for k in ("HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES", "ROCP_TOOL_LIBRARY", "ROCPROFILER_REGISTER_FORCE_LOAD") if os.environ.get(k)]
_TOOLS += [v for v in (os.environ.get("LD_PRELOAD", ""),) if "rocprof" in v or "roctracer" in v]
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(bool(_TOOLS), reason="a profiler's tool library is loaded (%s): the AQL queue is not taken" % _TOOLS)]
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold_tokens(name):
    return json.load(open(os.path.join(GOLD, name + ".json")))["argmax"]


@pytest.mark.parametrize("name,steps", [("tiny", 64), ("ragged", 33), ("tinylong", 1280), ("stories15M", 256), ("stories110M", 1024), ("llama2_7b_L2", 300)])
def test_greedy_loop_on_the_queue_equals_graph_replay_and_the_reference(name, steps):
    """Every step level a run crosses (one workgroup per head, the fused QKV + attention launch, eight workgroups per head), the
    latency and the streaming form, n % 4 != 0 shapes (scalar kernels): queue == graphs == the reference's tokens; a second run over
    the same context (the cache rows of the first run are still in somebody's L2), a run that starts in mid-context."""
    ctx = runtime.Context(configs.header(name))
    ctx.synth_fill(configs.DEFAULT_SEED)
    a = ctx.decode_greedy(1, 0, steps).tolist()
    assert ctx.get_option(runtime.OPT_AQL_QUEUE) == 1, runtime.lib().l2_last_error()
    want = gold_tokens(name)
    assert a == want[:steps]
    again = ctx.decode_greedy(1, 0, steps).tolist()
    assert again == a
    mid = steps // 3
    assert ctx.decode_greedy(a[mid - 1], mid, steps - mid).tolist() == a[mid:]
    # the pick is folded into the next token's first launch, the run's last one has a launch of its own: runs of one and two tokens
    assert ctx.decode_greedy(1, 0, 1).tolist() == a[:1] and ctx.decode_greedy(1, 0, 2).tolist() == a[:2]
    assert ctx.decode_greedy(a[4], 5, 1).tolist() == a[5:6]
    ctx.set_option(runtime.OPT_AQL_QUEUE, 0)
    assert ctx.get_option(runtime.OPT_AQL_QUEUE) == 0
    assert ctx.decode_greedy(1, 0, steps).tolist() == a
    assert ctx.decode_greedy(1, 0, 1).tolist() == a[:1] and ctx.decode_greedy(1, 0, 2).tolist() == a[:2]      # (the folded pick under graph replay)
    ctx.close()


@pytest.mark.parametrize("name", ["tiny", "stories15M", "stories110M"])
def test_blocking_call_on_the_queue_equals_graph_replay_bit_for_bit(name):
    """l2_forward through the queue ({token, pos} fetched from pinned host memory by the first launch, logits written straight into
    the host's buffer) against l2_forward through a replayed hipGraph: the same kernels in the same order -- logits bit for bit, at
    positions on both sides of every level change, with the greedy loop and prompt ingestion run in between (they leave their own
    lines in the caches the call must not trust)."""
    hdr = configs.header(name)
    q, g = runtime.Context(hdr), runtime.Context(hdr)
    for c in (q, g):
        c.synth_fill(configs.DEFAULT_SEED)
    g.set_option(runtime.OPT_AQL_QUEUE, 0)
    S = hdr[6]
    tok = 1
    for pos in range(min(S, 300)):
        a = np.array(q.forward(tok, pos), copy=True)
        b = g.forward(tok, pos)
        assert np.array_equal(a, b), (name, pos)
        tok = runtime.argmax(a)
        if pos == 40:      # the device loop in between, then the same position again
            assert q.decode_greedy(tok, pos + 1, 8).tolist() == g.decode_greedy(tok, pos + 1, 8).tolist()
    assert q.get_option(runtime.OPT_AQL_QUEUE) == 1 and g.get_option(runtime.OPT_AQL_QUEUE) == 0
    toks = gold_tokens(name)[:20]
    la, lb = q.prefill([1] + toks[:19], 0), g.prefill([1] + toks[:19], 0)
    assert np.array_equal(la, lb)
    assert np.array_equal(q.forward(toks[19], 20), g.forward(toks[19], 20))
    q.close(); g.close()


def test_fence_scopes_are_switches_not_requirements(monkeypatch):
    """The packets' fence scopes can be raised for A/B runs (L2_AQL_FENCE=1: what a hipGraph node carries); results do not change."""
    name, steps = "stories15M", 256
    want = gold_tokens(name)[:steps]
    for env in ({"L2_AQL_FENCE": "1"}, {"L2_AQL_FENCE": "2"}, {"L2_AQL_ACQ": "1", "L2_AQL_REL": "1", "L2_AQL_TOKACQ": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctx = runtime.Context(configs.header(name))
        ctx.synth_fill(configs.DEFAULT_SEED)
        assert ctx.decode_greedy(1, 0, steps).tolist() == want and ctx.get_option(runtime.OPT_AQL_QUEUE) == 1
        ctx.close()
        for k in env:
            monkeypatch.delenv(k)


def test_queue_follows_option_changes_and_uploads():
    """A recording holds the step's kernel arguments (pointers, shapes): whatever invalidates a captured graph invalidates it --
    kept-state reads switched on (other launches), a matrix uploaded again."""
    hdr = configs.header("stories15M")
    ctx = runtime.Context(hdr)
    ctx.synth_fill(configs.DEFAULT_SEED)
    a = ctx.decode_greedy(1, 0, 40).tolist()
    ctx.set_option(runtime.OPT_KEEP_STATE, 1)
    assert ctx.decode_greedy(1, 0, 40).tolist() == a
    assert np.abs(ctx.read_state("xb2")).max() > 0
    ctx.set_option(runtime.OPT_KEEP_STATE, 0)
    n = hdr[0] * hdr[0]
    ctx.upload(runtime.T_WO, 2, np.zeros(n, dtype=np.float32))
    b = ctx.decode_greedy(1, 0, 40).tolist()
    assert b != a
    other = runtime.Context(hdr)
    other.set_option(runtime.OPT_AQL_QUEUE, 0)
    other.synth_fill(configs.DEFAULT_SEED)
    other.upload(runtime.T_WO, 2, np.zeros(n, dtype=np.float32))
    assert other.decode_greedy(1, 0, 40).tolist() == b
    ctx.close(); other.close()


@pytest.mark.parametrize("name", ["stories15M", "stories110M"])
def test_sampled_loop_on_the_queue_equals_graph_replay(name):
    """l2_decode_sample through the queue (the sampler's launches handed over by its recorder hook, each acquiring at agent scope:
    its kernels are not under the coherence rule) against the same runs through replayed hipGraphs: tokens and RNG state, plain
    sample and top-p, across a level change, twice over the same context."""
    hdr = configs.header(name)
    q, g = runtime.Context(hdr), runtime.Context(hdr)
    for c in (q, g):
        c.synth_fill(configs.DEFAULT_SEED)
    g.set_option(runtime.OPT_AQL_QUEUE, 0)
    n = min(hdr[6], 300)
    for rep in range(2):
        for (t, p, seed) in ((0.9, 1.0, 42), (1.0, 0.9, 7), (0.7, 0.5, 1234567 + rep)):
            a, sa = q.decode_sample(1, 0, n, t, p, seed)
            b, sb = g.decode_sample(1, 0, n, t, p, seed)
            assert a.tolist() == b.tolist() and sa == sb, (name, t, p, rep)
    assert q.get_option(runtime.OPT_AQL_QUEUE) == 1 and g.get_option(runtime.OPT_AQL_QUEUE) == 0
    assert q.decode_greedy(1, 0, 50).tolist() == g.decode_greedy(1, 0, 50).tolist()      # (the greedy program beside the sampled ones)
    q.close(); g.close()


@pytest.mark.parametrize("zero_copy", ["1", "0"])
@pytest.mark.parametrize("graph", ["1", "0"])
def test_blocking_call_without_the_queue_under_direct_dispatch_off(tmp_path, graph, zero_copy):
    """Wherever the library's queue stands down (a profiler's tool library, L2_AQL=0, a tensor-parallel step with RCCL collectives) the
    blocking call (llama2.ts:468) runs as a replayed hipGraph or eager launches -- also under AMD_DIRECT_DISPATCH=0, a legitimate runtime
    setting under which round 5 saw it hand back another step's logits from the second token on (a stream copy of {token, pos} in front
    of the graph launch).  {token, pos} are now fetched inside the step: every token of the 256-step golden, in a fresh process started
    with that setting, with and without the host-mapped logits."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "call.py"
    script.write_text('''
import json, os, sys
sys.path.insert(0, %r)
import numpy as np
from llama2_ts_amd import configs, runtime
gold = json.load(open(os.path.join(%r, "tests", "golden", "stories15M.json")))["argmax"]
ctx = runtime.Context(configs.header("stories15M")); ctx.synth_fill(configs.DEFAULT_SEED)
tok, out = 1, []
for pos in range(256):
    tok = runtime.argmax(ctx.forward(tok, pos, view=True)); out.append(tok)
assert ctx.get_option(runtime.OPT_AQL_QUEUE) == 0, "the queue was in use: this run did not test the graph path"
bad = [i for i, (a, b) in enumerate(zip(out, gold)) if a != b]
print("first mismatch:", bad[:1], "dispatch:", ctx.dispatch_reason())
sys.exit(1 if bad else 0)
''' % (root, root))
    env = dict(os.environ, AMD_DIRECT_DISPATCH="0", L2_AQL="0", L2_USE_GRAPH=graph, L2_ZERO_COPY_LOGITS=zero_copy)
    r = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-1500:]


def test_dispatch_reason_and_the_position_check():
    """l2_dispatch_reason says why the library's queue is not in use without touching l2_last_error (which is for failures); L2_OPT_CHECK_POS
    makes the reference's calling convention -- pos = 0, 1, 2, ... (llama2.ts:464, 496) -- an enforced one: a position that skips ahead of
    the rows the cache holds is L2_E_STATE, re-feeding an earlier position or restarting at 0 is not."""
    ctx = runtime.Context(configs.header("tiny"))
    ctx.synth_fill(configs.DEFAULT_SEED)
    toks = ctx.decode_greedy(1, 0, 8).tolist()
    assert ctx.get_option(runtime.OPT_AQL_QUEUE) == 1 and ctx.dispatch_reason() == ""
    ctx.set_option(runtime.OPT_AQL_QUEUE, 0)
    assert "switched off" in ctx.dispatch_reason()
    ctx.set_option(runtime.OPT_AQL_QUEUE, 1)
    ctx.set_option(runtime.OPT_USE_GRAPH, 0)
    assert "not recorded" in ctx.dispatch_reason() and ctx.decode_greedy(1, 0, 8).tolist() == toks
    ctx.set_option(runtime.OPT_USE_GRAPH, 1)
    assert ctx.dispatch_reason() == "" and ctx.decode_greedy(1, 0, 8).tolist() == toks
    # positions
    assert ctx.get_option(runtime.OPT_CHECK_POS) == 0
    ctx.forward(1, 20)                                     # default: any position is accepted (rows 8 .. 19 are whatever the cache holds)
    ctx.set_option(runtime.OPT_CHECK_POS, 1)
    ctx.forward(1, 0)                                      # a restart
    for pos in range(1, 6):
        ctx.forward(toks[pos - 1], pos)
    ctx.forward(toks[2], 3)                                # re-feeding an earlier position
    with pytest.raises(runtime.L2Error) as e:
        ctx.forward(toks[3], 6)                            # rows 0 .. 3 are written: 6 skips row 4 and 5 of THIS sequence
    assert e.value.code == -4 and "skips ahead" in str(e.value)
    ctx.forward(toks[3], 4)
    assert ctx.decode_greedy(toks[4], 5, 10).tolist()      # the device loop writes rows 5 .. 14
    ctx.forward(1, 15)
    with pytest.raises(runtime.L2Error):
        ctx.prefill([1, 2, 3], 30)
    ctx.prefill([1, 2, 3], 16)
    ctx.close()


def test_a_failed_run_retires_the_queue_and_the_context_goes_on_with_graphs(monkeypatch):
    """aql_run gives a run up when the queue makes no progress for L2_QUEUE_WAIT_S (or the runtime reports a queue error): the call fails, the
    queue -- whose ring may still hold the run's packets -- is destroyed with its recordings, and the context goes on with replayed hipGraphs.
    The hook makes the third run on the queue REPORT a failure after it completed (no GPU is hung for a test): that call raises, the next ones
    decode the reference's tokens through graphs, and l2_dispatch_reason says why (l2_last_error is left to the failure)."""
    monkeypatch.setenv("L2_DEBUG_FAIL_AQL_RUN", "3")
    name = "stories15M"
    want = gold_tokens(name)
    ctx = runtime.Context(configs.header(name))
    ctx.synth_fill(configs.DEFAULT_SEED)
    assert ctx.decode_greedy(1, 0, 64).tolist() == want[:64] and ctx.get_option(runtime.OPT_AQL_QUEUE) == 1      # run 1
    assert runtime.argmax(ctx.forward(1, 0)) == want[0]                                                          # run 2 (the blocking call)
    with pytest.raises(runtime.L2Error) as e:
        ctx.decode_greedy(1, 0, 64)                                                                              # run 3: reported as failed
    assert e.value.code == -3 and "AQL queue" in str(e.value)
    assert ctx.get_option(runtime.OPT_AQL_QUEUE) == 0 and "given up after a failed run" in ctx.dispatch_reason()
    assert ctx.decode_greedy(1, 0, 256).tolist() == want[:256]                                                   # hipGraphs from here on
    tok = 1
    for pos in range(16):
        tok = runtime.argmax(ctx.forward(tok, pos))
        assert tok == want[pos]
    toks, _ = ctx.decode_sample(1, 0, 32, 0.0, 1.0, 5)
    assert toks.tolist() == want[:32]
    ctx.close()
