"""Tensor-parallel shard layout on one GPU (no communicator): every rank's slices, whether generated on the
device or cut out of full host arrays by l2_upload, must equal llama2_ts_amd.tp.tensor_slice of the oracle's
tensors.  RCCL with more than one rank needs >1 GPU (bench.py --gpus N on the driver's node); here the RCCL calls
run with a 1-rank communicator, and the complete G-rank step runs through the library's loopback test hook
(G contexts, G host threads, one device).  The collectives' arithmetic is also covered over gloo in test_tp_gloo.py."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from llama2_ts_amd import configs, runtime, tp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

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


@pytest.mark.parametrize("collective", ["rccl", "p2p"])
def test_one_rank_communicator(collective):
    """L2_TP_FORCE_COMM=1: a 1-rank RCCL communicator drives the tensor-parallel code path on a single GPU --
    `rccl`: fp64 partials, ncclAllReduce(double, sum), residual kernel, ncclAllGather of the logits, captured into the
    per-token graph (or eager launches where RCCL refuses capture);
    `p2p`: IPC handles through ncclAllGather, the self-test, then the one-shot exchange kernels inside one captured
    graph per token.  Results must match the goldens of the TRUE reference like the ordinary path does."""
    import json
    meta = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tiny.json")))
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "tiny.npz"))
    os.environ["L2_TP_FORCE_COMM"] = "1"
    os.environ["L2_TP_ALLREDUCE"] = collective
    try:
        ctx = runtime.Context(meta["header"])
        # rccl: the collectives are captured into the per-token graph when this RCCL allows stream capture (mode 2), else launched eagerly (mode 1)
        assert ctx.tp_mode_id() == 3 if collective == "p2p" else ctx.tp_mode_id() in (1, 2)
        print("\n[one-rank communicator, %s] %s" % (collective, ctx.tp_mode()))
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
        del os.environ["L2_TP_ALLREDUCE"]


# a vocabulary (shard) of FEWER 256-element blocks than dim has: the logits gather then runs fewer workgroups than the combine launch of
# the pushed all-reduce, and the two must not share exchange counters (round-5 advisor finding: the combine's upper blocks fell one
# exchange behind the pushing GEMV, which reads block 0's counter -- the creation soak then sent the group back to RCCL, or a wait ran
# into its bound).  Llama-2-7B never showed it: 4000 / 256 and 4096 / 256 both round up to 16 blocks.
NARROW_VOCAB = (1024, 1536, 2, 8, 8, -512, 40)


def test_pushed_exchange_with_fewer_vocabulary_blocks_than_dim():
    orc = O.Oracle(NARROW_VOCAB, 7)
    want, tok = [], 1
    for pos in range(12):
        lg = orc.forward(tok, pos)
        want.append(lg)
        tok = O.argmax(lg)
    picks = [O.argmax(lg) for lg in want]
    os.environ["L2_TP_FORCE_COMM"] = "1"
    os.environ["L2_TP_ALLREDUCE"] = "p2p"
    os.environ["L2_TP_WAIT_S"] = "3"
    try:
        ctx = runtime.Context(NARROW_VOCAB)
        assert ctx.tp_mode_id() == 3, ctx.tp_mode()          # the creation soak (96 all-reduces, 32 gathers in between) kept the peer-to-peer path
        ctx.synth_fill(7)
        tok = 1
        for pos in range(12):
            got = np.array(ctx.forward(tok, pos), copy=True)
            assert np.abs(got - want[pos]).max() <= 1e-4, pos
            tok = picks[pos]
        assert ctx.decode_greedy(1, 0, 12).tolist() == picks
        ctx.close()
        # ... and a shard-timing context of an 8-rank group of the shape the finding names (d = 5120, V = 32000: 16 gather blocks, 20 combine
        # blocks): no soak runs there, a lagging block would sit out L2_TP_WAIT_S on every exchange and break the context
        solo = runtime.Context((5120, 13824, 1, 40, 40, -32000, 16), tp_rank=0, tp_size=8, nccl_id=runtime.TP_SOLO_ID)
        solo.synth_fill(1)
        solo.decode_greedy(1, 0, 12)
        solo.decode_greedy(1, 0, 12)
        solo.close()
    finally:
        for k in ("L2_TP_FORCE_COMM", "L2_TP_ALLREDUCE", "L2_TP_WAIT_S"):
            del os.environ[k]


def _run_group(name, G, n_forward, n_greedy, exact=False, collective="p2p"):
    """G ranks of one tensor-parallel group as G host threads on one device (L2_TP_LOOPBACK test hook in
    llama2_hip.hip: the collectives become device sums/copies between thread barriers; everything else is the
    code the RCCL path runs).  Returns per-rank (logits[n_forward][V], greedy tokens)."""
    import json
    import threading
    here = os.path.dirname(__file__)
    meta = json.load(open(os.path.join(here, "golden", name + ".json")))
    gid = bytes([G, len(name)] + [7] * 126)
    out, errs = [None] * G, [None] * G

    def rank_main(r):
        try:
            ctx = runtime.Context(meta["header"], tp_rank=r, tp_size=G, nccl_id=gid)
            ctx.synth_fill(meta["seed"])
            if exact:
                ctx.set_option(1, 1)
            logits = [np.array(ctx.forward(tok, pos), copy=True) for pos, tok in enumerate(meta["tokens_fed"][:n_forward])]
            toks = ctx.decode_greedy(meta["tokens_fed"][0], 0, n_greedy).tolist() if n_greedy else []
            out[r] = (np.stack(logits), toks)
            ctx.close()
        except BaseException as e:   # surfaced by the main thread
            errs[r] = e

    os.environ["L2_TP_LOOPBACK"] = "1"
    os.environ["L2_TP_ALLREDUCE"] = collective   # p2p: the exchange kernels; rccl: host-side stand-ins of the two collectives
    try:
        threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(G)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(180)
        assert not any(t.is_alive() for t in threads), "a rank hung"
    finally:
        del os.environ["L2_TP_LOOPBACK"]
        del os.environ["L2_TP_ALLREDUCE"]
    for e in errs:
        if e is not None:
            raise e
    return meta, out


@pytest.mark.parametrize("collective", ["p2p", "rccl"])
@pytest.mark.parametrize("name,G,steps", [("tiny", 2, 24), ("tiny", 4, 24), ("stories15M", 2, 16),
                                          ("llama2_7b_L2", 4, 6), ("llama2_7b_L2", 8, 6)])
def test_tensor_parallel_group_matches_reference(name, G, steps, collective):
    """The whole tensor-parallel step with G > 1 ranks -- row / column slices, fp64 partials of wo and w2 summed
    across ranks and rounded once, logits gathered, greedy loop on the gathered logits -- against the goldens of
    the TRUE reference.  Every rank must hold the same full logits.  `p2p`: the one-shot peer-to-peer exchange
    kernels (the ranks' inboxes are ordinary device pointers here, peer-mapped over xGMI on a real node; with all
    ranks on one GPU the contribute and combine halves of an exchange are two launches around a host barrier --
    kernels that wait for each other need one GPU per rank -- the single-kernel form runs in test_one_rank_communicator
    and test_two_gpu_group_over_rccl_and_xgmi); `rccl`: the loopback stand-ins of ncclAllReduce / ncclAllGather between host barriers."""
    meta, out = _run_group(name, G, steps, steps, collective=collective)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    for r in range(G):
        logits, toks = out[r]
        kept = meta.get("logit_positions") or list(range(len(g["logits"])))    # big vocabularies keep a few positions
        for row, pos in enumerate(kept):
            if pos < steps:
                err = float(np.abs(logits[pos] - g["logits"][row]).max())
                assert err <= 1e-4, (r, pos, err)
        assert [int(runtime.argmax(row)) for row in logits] == meta["argmax"][:steps]
        assert np.array_equal(logits, out[0][0])          # identical on every rank
        if meta["tokens_fed"][:steps] == [1] + meta["argmax"][:steps - 1]:   # the golden run was greedy from BOS
            assert toks == meta["argmax"][:steps]


def test_full_llama2_7b_as_an_eight_rank_group_matches_the_reference():
    """The shape the tensor-parallel path exists for, all 32 layers: eight ranks (host threads on the one GPU, 3.4 GB of shards each) decode the
    first 64 positions of the FULL Llama-2-7B golden -- the real reference's own 47-minute run (tests/golden/llama2_7b.json) -- through the
    one-shot peer-to-peer exchange: every argmax, the logits at the kept positions 0, 2, 19 and 63 within 1e-4, identical on every rank, and the
    device loop's tokens (llama2.ts:270, 292 are the reduce points: 4 160 exchanges per run here)."""
    steps = 64
    meta, out = _run_group("llama2_7b", 8, steps, steps, collective="p2p")
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "llama2_7b.npz"))
    for r in range(8):
        logits, toks = out[r]
        for row, pos in enumerate(meta["logit_positions"]):
            if pos < steps:
                assert float(np.abs(logits[pos] - g["logits"][row]).max()) <= 1e-4, (r, pos)
        assert [int(runtime.argmax(row)) for row in logits] == meta["argmax"][:steps]
        assert np.array_equal(logits, out[0][0]) and toks == meta["argmax"][:steps]


def test_tensor_parallel_group_samples_like_a_single_rank():
    """l2_decode_sample on a 2-rank group: every rank samples from the gathered logits with the same seed, so all ranks
    and the single-GPU context must produce the same token ids and the same advanced RNG state."""
    import threading
    hdr = configs.header("tiny")
    single = runtime.Context(hdr)
    single.synth_fill(3)
    want, want_rng = single.decode_sample(1, 0, 20, 0.8, 0.9, 1234)
    single.close()
    G, gid = 2, bytes([9] * 128)
    out, errs = [None] * G, [None] * G

    def rank_main(r):
        try:
            ctx = runtime.Context(hdr, tp_rank=r, tp_size=G, nccl_id=gid)
            ctx.synth_fill(3)
            out[r] = ctx.decode_sample(1, 0, 20, 0.8, 0.9, 1234)
            ctx.close()
        except BaseException as e:
            errs[r] = e

    os.environ["L2_TP_LOOPBACK"] = "1"
    try:
        ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(G)]
        [t.start() for t in ts]
        [t.join(120) for t in ts]
        assert not any(t.is_alive() for t in ts)
    finally:
        del os.environ["L2_TP_LOOPBACK"]
    for e in errs:
        if e is not None:
            raise e
    for r in range(G):
        assert out[r][0].tolist() == want.tolist() and out[r][1] == want_rng


# The 4- and 8-process groups are box-dependent stand-ins (profiles/r05/tp_process_group_one_gpu_flakiness.txt: some boxes never keep the
# kernels of 8 processes resident together, and three bounded-wait attempts cost the round-5 driver run 13 of its 16 minutes).  The
# default `-m gpu` run keeps the 2-process group, which must pass and covers the mechanism; L2_TEST_WIDE_PROCESS_GROUPS=1 adds 4 and 8.
_WIDE_GROUPS = os.environ.get("L2_TEST_WIDE_PROCESS_GROUPS", "0") == "1"


@pytest.mark.parametrize("G", [2] + ([4, 8] if _WIDE_GROUPS else []))
def test_process_group_on_one_gpu_through_ipc(tmp_path, G):
    """What one GPU can say about the multi-GPU path: the G ranks as separate PROCESSES (started fresh), meeting through files
    (L2_TP_IPC_DIR -- RCCL refuses two ranks on one device), each mapping the other's uncached inboxes with
    hipIpcOpenMemHandle, the start-up self-test, then the peer-to-peer all-reduce / gather inside one hipGraph per token
    between the processes: logits against the goldens of the TRUE reference on every rank.  What stays untested is
    the same mapping ACROSS GPUs (xGMI peer access) and RCCL with more than one rank."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    meet = tmp_path / "meet"
    meet.mkdir()
    script = tmp_path / "rank.py"
    script.write_text('''
import json, os, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
from llama2_ts_amd import runtime
rank, G = int(sys.argv[1]), int(sys.argv[2])
meta = json.load(open(os.path.join(%r, "tests", "golden", "llama2_7b_L2.json")))
g = np.load(os.path.join(%r, "tests", "golden", "llama2_7b_L2.npz"))
ctx = runtime.Context(meta["header"], device=0, tp_rank=rank, tp_size=G, nccl_id=b"x" * 128)
print("rank", rank, "mode:", ctx.tp_mode(), flush=True)
ctx.synth_fill(meta["seed"])
keep = {p: i for i, p in enumerate(meta["logit_positions"])}
for pos, tok in enumerate(meta["tokens_fed"][:40]):
    lg = ctx.forward(tok, pos)
    assert runtime.argmax(lg) == meta["argmax"][pos], (rank, pos)
    if pos in keep:
        assert np.abs(lg - g["logits"][keep[pos]]).max() <= 1e-4, (rank, pos)
assert ctx.decode_greedy(1, 0, 40).tolist() == meta["argmax"][:40]
ctx.close()
print("rank", rank, "ok", flush=True)
''' % (root, root, root, root))
    def run_group(meet_dir):
        env = dict(os.environ, L2_TP_IPC_DIR=str(meet_dir), HSA_ENABLE_IPC_MODE_LEGACY="0", L2_TP_WAIT_S="10")
        procs = [subprocess.Popen([sys.executable, str(script), str(r), str(G)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(G)]
        outs = []
        for pr in procs:
            try:
                outs.append(pr.communicate(timeout=600)[0].decode())
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
        bad = [out[-3000:] for r, (pr, out) in enumerate(zip(procs, outs)) if pr.returncode != 0 or "rank %d ok" % r not in out or "peer-to-peer" not in out]
        return bad

    # G kernels of G PROCESSES that wait for each other's flags are not guaranteed to be resident together on ONE GPU (the
    # product runs one rank per GPU): if the GPU's process scheduler serialises them a bounded wait gives up (seen once in several
    # hundred runs, on a freshly started box).  That is a property of this one-GPU stand-in, so the group gets a second try.
    # Only THAT failure is retried: wrong tokens, a golden mismatch or a missing marker fail at once.
    # Round 5: on some boxes of the pool eight processes on one GPU fail this way on every attempt -- with this round's library and with
    # the rounds before it alike (profiles/r05/tp_process_group_one_gpu_flakiness.txt): the scheduler runs them one after the other,
    # a rank's kernel is off the chip for longer than the bound of a wait (also the 2 s bound of the hand-off inside the fused
    # attention + wo launch).  Three attempts; a group of four or eight processes that still cannot be kept resident is SKIPPED
    # with that reason (two processes must pass), anything else fails.
    markers = ("never raised its flag", "did not arrive", "failed its self-test", "exchange timed out", "hand-off granule inside a fused launch never arrived")
    # (round 6: ONE rank whose bounded wait gave up poisons the attempt for all of them -- it goes on pushing rows of a step whose results are
    # invalid, with valid tags, so its peers add garbage and fail their golden check without any error of their own: an attempt in which ANY rank
    # reports a bounded wait says nothing about the others)
    def co_residency(outputs):
        return bool(outputs) and any(any(m in b for m in markers) for b in outputs)
    bad = run_group(meet)
    for attempt in (2, 3):
        if not co_residency(bad):
            break
        import warnings
        warnings.warn("process group of %d on one GPU: a bounded wait gave up (co-residency); attempt %d" % (G, attempt))
        again = tmp_path / ("meet%d" % attempt)
        again.mkdir()
        bad = run_group(again)
    if G >= 4 and co_residency(bad):
        pytest.skip("this box does not keep the kernels of %d processes resident together on its one GPU: bounded waits gave up on three attempts "
                    "(a property of the one-GPU stand-in -- the product runs one rank per GPU; the 2-process group, which must pass, covers the mechanism)" % G)
    assert not bad, bad[0]


def test_two_gpu_group_over_rccl_and_xgmi(tmp_path):
    """The real thing, on a box with at least two GPUs (skipped on the 1-GPU development boxes): two PROCESSES, one
    per GPU, started fresh (nothing in this process's GPU state is inherited), rendezvous over gloo on 127.0.0.1,
    l2_create_tp with a real ncclUniqueId; each rank checks its logits against the goldens of the TRUE reference for
    both collectives (RCCL, and the peer-to-peer exchange over IPC-mapped inboxes)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rank.py"
    script.write_text('''
import json, os, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import torch, torch.distributed as dist
from llama2_ts_amd import runtime
import ctypes as C
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
meta = json.load(open(os.path.join(%r, "tests", "golden", "llama2_7b_L2.json")))
g = np.load(os.path.join(%r, "tests", "golden", "llama2_7b_L2.npz"))
idbuf = torch.zeros(128, dtype=torch.uint8)
if rank == 0:
    b = C.create_string_buffer(128)
    runtime._check(runtime.lib().l2_tp_unique_id(b))
    idbuf = torch.frombuffer(bytearray(b.raw), dtype=torch.uint8).clone()
dist.broadcast(idbuf, 0)
ctx = runtime.Context(meta["header"], device=int(os.environ["LOCAL_RANK"]), tp_rank=rank, tp_size=world, nccl_id=bytes(idbuf.numpy().tobytes()))
print("rank", rank, "mode:", ctx.tp_mode(), flush=True)
ctx.synth_fill(meta["seed"])
keep = {p: i for i, p in enumerate(meta["logit_positions"])}
for pos, tok in enumerate(meta["tokens_fed"][:66]):
    lg = ctx.forward(tok, pos)
    assert runtime.argmax(lg) == meta["argmax"][pos], (rank, pos)
    if pos in keep:
        assert np.abs(lg - g["logits"][keep[pos]]).max() <= 1e-4, (rank, pos)
assert ctx.decode_greedy(1, 0, 66).tolist() == meta["argmax"][:66]
ctx.close()
dist.barrier()
''' % (root, root, root, root))
    for collective in ("rccl", "p2p"):
        env = dict(os.environ, L2_TP_ALLREDUCE=collective, HSA_ENABLE_IPC_MODE_LEGACY="0")
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", "29517", str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        out = r.stdout.decode("utf8", "replace")
        assert r.returncode == 0, out[-3000:]
        assert ("peer-to-peer" in out) == (collective == "p2p"), out[-2000:]


def test_context_on_a_device_other_than_the_threads_current_one():
    """A context is bound to the device it was created on, whatever the calling thread has current: the first step allocates and
    builds the repacked copies (ensure_ready), and that has to happen on the context's device -- one process holding two contexts
    on two GPUs, or an N-API worker thread.  Needs two GPUs (skipped on the 1-GPU development boxes)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "llama2_7b_L2.json")))
    ctx = runtime.Context(meta["header"], device=1)
    ctx.synth_fill(meta["seed"])
    torch.cuda.set_device(0)                      # the thread's current device is NOT the context's
    free0 = torch.cuda.mem_get_info(0)[0]
    toks = ctx.decode_greedy(1, 0, 8).tolist()    # packs (2 GB of repacked copies) and runs on device 1
    assert toks == meta["argmax"][:8]
    assert ctx.get_option(runtime.OPT_PACKED_MIB) > 1000
    assert free0 - torch.cuda.mem_get_info(0)[0] < (256 << 20)       # nothing of it landed on device 0
    ctx.close()
