"""bench.py without a GPU: the launcher half of `--gpus N` and the honesty of the failure path.  On a box with no gfx950 device
the ranks cannot create a context, so there is NO result line and the exit code is not 0 -- the benchmark never prints a number
it did not measure (there is no CPU fallback behind it)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


@pytest.mark.skipif(not _no_gpu(), reason="a GPU is visible here: tests/test_bench_gpu.py covers the launcher with real ranks")
def test_gpus_flag_starts_ranks_and_relays_their_failure():
    """No GPU: every way of running the ranks fails at once (no HIP device), stage after stage, each from fresh workers.  The line still
    appears -- and says that nothing was measured: value null, `failed`, the note of every stage; the exit code is not 0."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "llama2_7b_L2", "--steps", "4", "--warmup", "1",
                        "--no-cpu-baseline"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=env)
    assert r.returncode != 0
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["value"] is None and j["failed"] is True and j["n_gpus"] == 2          # no number was invented
    for label in ("RCCL + peer-to-peer exchange:", "RCCL collectives only:", "file rendezvous + peer-to-peer exchange:", "independent replicas:"):
        assert label in j["note"], j["note"]
    assert "no HIP device" in j["note"] or "no HIP device" in r.stderr.decode()
    assert [f["stage"] for f in j["tp"]["formation"]] == ["p2p", "rccl", "file", "replicas"] and not any(f["ok"] for f in j["tp"]["formation"])


STUB = os.path.join(ROOT, "tests", "stub", "bench_worker_stub.py")


def _supervised(tmp_path, hang="", fail="", launcher=True, deadlines="4,4,4,20", port=29731):
    """`bench.py --gpus 2 --config llama2_7b_L2` with the rank workers replaced by tests/stub/bench_worker_stub.py (no GPU needed): through
    torch.distributed.run as the driver starts it (`launcher`), or through bench.py's own launcher."""
    pids = tmp_path / "pids.txt"
    env = dict(os.environ, L2_TEST_HOOKS="1", L2_BENCH_WORKER_STUB=STUB, L2_STUB_HANG=hang, L2_STUB_FAIL=fail, L2_STUB_PIDS=str(pids), L2_BENCH_STAGE_DEADLINES=deadlines)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "L2_TP_IPC_DIR"):
        env.pop(k, None)
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "llama2_7b_L2", "--steps", "4", "--warmup", "1"]
    cmd = ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port)] if launcher else [sys.executable]) + tail
    import time
    t0 = time.monotonic()
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=env)
    took = time.monotonic() - t0
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, (r.stdout.decode()[-1500:], r.stderr.decode()[-1500:])
    started = [ln.split() for ln in pids.read_text().splitlines()]
    for pid, _stage, _rank in started:      # no worker outlives the run: the hung ones were killed with their process groups
        assert not os.path.exists("/proc/%s" % pid) or open("/proc/%s/stat" % pid).read().split()[2] == "Z", ("worker %s still alive" % pid)
    return r.returncode, json.loads(lines[0]), took, started, r.stderr.decode()


def test_a_rank_that_hangs_costs_a_deadline_not_the_line(tmp_path):
    """The first multi-GPU contact (llama2.ts:270, 292 are the all-reduce points): RCCL has no timeout, so a rank may HANG, not fail.  Stub
    ranks that sleep forever in l2_create_tp (p2p), before they even start (rccl) and inside the proof decode (file): each stage is given
    up at its deadline, its workers are killed, the next stage starts FRESH processes, and the replicas stage delivers the line --
    with rc 0, the note of every stage, `tp.formation`, roofline and cpu_baseline."""
    rc, j, took, started, err = _supervised(tmp_path, hang="p2p:create,rccl:start,file:prove")
    assert rc == 0 and j["value"] == 123.0 and j["config"]["parallelism"] == "replicas2" and j["scaling"] == "weak", (j, err[-800:])
    assert took < 90, took
    note = j["note"]
    assert "RCCL + peer-to-peer exchange: rank" in note and "phase 'create'" in note and "RCCL collectives only: rank" in note and "phase 'start'" in note
    assert "file rendezvous + peer-to-peer exchange: rank" in note and "phase 'prove'" in note and "measured 2 independent replicas instead" in note
    f = j["tp"]["formation"]
    assert [x["stage"] for x in f] == ["p2p", "rccl", "file", "replicas"] and [x["ok"] for x in f] == [False, False, False, True]
    assert f[0]["timed_out_ranks"] == [0, 1] and "roofline" in j and "cpu_baseline" in j
    assert sorted((s, r) for _p, s, r in started) == sorted((s, str(r)) for s in ("p2p", "rccl", "file", "replicas") for r in (0, 1))      # a fresh worker per stage and rank
    assert len({p for p, _s, _r in started}) == 8


def test_every_stage_hanging_still_ends_with_a_line_and_a_nonzero_exit(tmp_path):
    rc, j, took, _started, _err = _supervised(tmp_path, hang="p2p:create,rccl:create,file:create,replicas:run", deadlines="4,3,3,5", port=29733)
    assert rc != 0 and j["value"] is None and j["failed"] is True and took < 90
    assert j["note"].count("no progress within") == 4 and "independent replicas: rank" in j["note"]


def test_one_rank_failing_stops_the_others_waiting(tmp_path):
    """Rank 1 cannot form the group in the p2p stage while rank 0 sits in its create (as it would inside a collective its peer never
    joins): rank 0's supervisor hears about it through the launcher's store and does not sit out the 60 s deadline.  The rccl stage then works."""
    rc, j, took, _started, _err = _supervised(tmp_path, hang="p2p:create", fail="p2p:1", deadlines="20,60,20,20", port=29735)
    assert rc == 0 and j["config"]["parallelism"] == "tp2" and j["tp"]["stage"] == "rccl" and j["tp"]["env"]["L2_TP_ALLREDUCE"] == "rccl"
    assert took < 45, took
    assert "RCCL + peer-to-peer exchange: rank 1: L2Error: stub rank 1 cannot form the group in stage p2p" in j["note"]


def test_the_script_supervises_its_own_launcher(tmp_path):
    """`python bench.py --gpus 2` (no launcher, what tools/first_contact.sh runs): the same supervision behind bench.py's own child launcher;
    the file stage gets a meeting directory and the narrow switch, and only that stage does."""
    rc, j, took, _started, _err = _supervised(tmp_path, hang="p2p:create,rccl:prove", launcher=False)
    assert rc == 0 and j["tp"]["stage"] == "file" and j["config"]["parallelism"] == "tp2" and took < 90
    assert j["tp"]["env"]["L2_TP_FILE_RENDEZVOUS"] == "1" and os.path.basename(j["tp"]["env"]["L2_TP_IPC_DIR"]).startswith("l2_meet_")
    assert "the ranks met through files and exchange peer to peer (no RCCL)" in j["note"]


@pytest.mark.skipif(not _no_gpu(), reason="a GPU is visible here")
def test_single_process_fails_loudly_without_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "stories15M", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-pmc"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0 and b"no HIP device visible" in r.stderr
    assert not [l for l in r.stdout.decode().splitlines() if l.startswith("{")]


def test_reference_js_figures_are_fixtures_with_provenance():
    """cpu_baseline.reference_js_tok_s comes from tests/golden/reference_speed.json (oracle/make_goldens.py --speed): every entry says
    which CPU / Node / argv produced it."""
    sys.path.insert(0, ROOT)
    import bench
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_speed.json")))
    for name in ("stories15M", "stories110M", "llama2_7b"):
        assert ref[name]["tok_s"] > 0 and ref[name]["threads"] == 1 and ref[name]["node"].startswith("v") and ref[name]["cpu"]
        got = bench.reference_js_figure(name)
        assert got["reference_js_tok_s"] == round(ref[name]["tok_s"], 4) and "build container" in got["reference_js_measured"]
    assert bench.reference_js_figure("tiny") == {"reference_js_tok_s": None}
