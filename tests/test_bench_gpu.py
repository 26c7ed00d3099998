"""bench.py contract on a real GPU: one JSON line with the driver's keys, the roofline and cpu_baseline objects,
and the multi-process launch path (gloo rendezvous, barriers, max-over-ranks) with two replica ranks on one GPU."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "stories15M", "--steps", "32", "--warmup", "4"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 32 and j["warmup"] == 4 and j["unit"] == "tokens/s" and j["vs_baseline"] is None
    assert "workload" in j["config"] and "model" not in j["config"]
    # the dispatch the timed run used (csrc/aql_queue.h): the library's own queue -- or replayed hipGraphs WITH the reason (e.g. under a profiler's tool library)
    loop = j["config"]["loop"]
    assert "AQL packets on the library's own HSA queue" in loop or ("one hipGraph replay per token (" in loop and "AQL" in loop) or "eager launches (" in loop, loop
    wm = j["config"]["weights_mib"]
    assert wm["repacked"] == 0 and abs(wm["on_device"] - wm["checkpoint"]) <= 2          # stories15M: latency-form phases, a one-batch classifier
    rf = j["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and "sample" in cb
    import shutil
    if shutil.which("node"):      # the reference's arithmetic in the reference's runtime, on this box (oracle/llama2_oracle.mjs)
        jp = cb["js_port"]
        assert jp["value"] > 0 and jp["cores"] == 1 and jp["tokens_equal_reference_golden"] is True and "node" in jp["runtime"]
    assert abs(j["value"] - 1e3 / j["ms_per_step"]) / j["value"] < 0.01
    # the run that was TIMED is checked against the real reference's tokens (tests/golden/stories15M.json), and so is the drop-in loop
    pr = j["parity"]
    assert pr["steps_checked"] == 32 and pr["equal_to_reference_golden"] is True and pr["dropin_equal_to_reference_golden"] is True
    assert rf["kernel_trace_us"] is None or 0 < rf["kernel_trace_us"] <= rf["avg_launch_us"] * 1.2
    assert "duration_used" in rf
    assert rf["mfma_insts"] in (0, None), rf      # the decode path issues no MFMA: a counter pass of this run (None: no rocprofv3 here)
    # `value` is the device-resident loop (a SURVEY.md 8(f1) extra) and says so; the contract's own call -- one blocking transformer() per
    # token, logits to the host -- stands next to it: through the N-API addon where Node is here, else through ctypes
    assert "`value` times THIS loop" in loop and "contract_tok_s" in loop
    assert 0 < j["contract_tok_s"] <= j["value"] * 1.02 and 0 < j["contract_hbm_frac"] < 1 and "contract_how" in j
    napi = (j.get("napi_dropin_tok_s") or {}).get("value")
    assert j["contract_tok_s"] == (napi or j["dropin_tok_s"])
    assert abs(j["contract_hbm_frac"] - j["algorithmic_bytes_per_token"] * j["contract_tok_s"] / 8e12) < 2e-4


def test_bench_exits_nonzero_when_the_timed_tokens_are_not_the_references(tmp_path):
    """A benchmark that decodes other tokens than the reference did is not a measurement: same shape, another seed for the weights,
    the golden file of the default seed forged to claim that seed -> the line carries the first mismatch and the exit code is 3."""
    import shutil
    work = tmp_path / "repo"
    shutil.copytree(ROOT, work, ignore=shutil.ignore_patterns(".git", "gpurun_out", "profiles", "__pycache__", "*.npz", "tools"))
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "stories15M.json")))
    g["seed"] = g["seed"] + 1
    json.dump(g, open(work / "tests" / "golden" / "stories15M.json", "w"))
    r = subprocess.run([sys.executable, str(work / "bench.py"), "--config", "stories15M", "--steps", "16", "--warmup", "2", "--seed", str(g["seed"]),
                        "--no-cpu-baseline", "--no-pmc", "--no-extra", "--no-dropin"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 3, (r.returncode, r.stderr.decode()[-1500:])
    j = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][0])
    assert j["parity"]["equal_to_reference_golden"] is False and "first_mismatch" in j["parity"]
    assert "PARITY FAILURE" in r.stderr.decode()


def test_bench_two_ranks_replicas_on_one_gpu():
    env = dict(os.environ, L2_BENCH_FORCE_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29611", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "16", "--warmup", "2",
           "--config", "stories15M", "--no-cpu-baseline"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1                       # rank 0 only
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["config"]["parallelism"] == "replicas2"
    assert j["value"] > 0
    # every rank was a CPU-only supervisor with one fresh worker: one stage (a shape that does not shard), proved before it was timed
    assert [f["stage"] for f in j["tp"]["formation"]] == ["replicas"] and j["tp"]["formation"][0]["ok"] and j["tp"]["ranks"] == 2
    assert j["tp"]["proved_before_timing"]["same_on_every_rank"] and j["tp"]["proved_before_timing"]["equals_reference_golden"] is True
    assert j["roofline"]["bound"] == "hbm" and j["roofline"]["frac"] > 0 and "cpu_baseline" in j


CO_RESIDENCY = ("never raised its flag", "did not arrive", "failed its self-test", "exchange timed out")


def _retry_once(fn):
    """Two ranks' exchange kernels on ONE GPU wait for each other without a guarantee of being resident together (the product runs one
    rank per GPU); a BOUNDED WAIT that gives up there says nothing about the code under test, so these stand-ins get a second try --
    for that failure only: wrong tokens, a golden mismatch or a missing marker fail at once (a race in the flag protocol must not
    pass on its second run)."""
    try:
        return fn(0)
    except AssertionError as e:
        if not any(m in str(e) for m in CO_RESIDENCY):
            raise
        import warnings
        warnings.warn("two ranks on one GPU: a bounded wait gave up on the first attempt (co-residency); retried once: %s" % str(e)[:300])
        return fn(1)


def _two_ranks(extra_env, tmp_path, port):
    env = dict(os.environ, L2_BENCH_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "16", "--warmup", "2",
           "--config", "llama2_7b_L2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_tensor_parallel_flow_with_two_processes_on_one_gpu(tmp_path):
    """The line `bench.py --gpus 2` prints for the 7B shape, with its two ranks as processes on the one GPU that meet
    through files (L2_TP_IPC_DIR): strong scaling, the tensor-parallel step named, the peer-to-peer exchange used."""
    def attempt(k):
        meet = tmp_path / ("meet%d" % k)
        meet.mkdir()
        j = _two_ranks({"L2_TP_IPC_DIR": str(meet), "L2_TP_WAIT_S": "10"}, tmp_path, 29613 + 20 * k)
        assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["parallelism"] == "tp2" and j["value"] > 0, j.get("note")
        assert "peer-to-peer" in j["config"]["loop"] and "note" not in j, j.get("note")
    _retry_once(attempt)


def test_bench_second_chance_without_rccl(tmp_path):
    """Two ranks on one device over RCCL: ncclCommInitRank refuses the duplicate GPU, every rank hears about it over gloo,
    the job tries RCCL collectives only (refused again), then lets the ranks meet through files (no RCCL): tensor parallel after
    all, with a note that names every step."""
    def attempt(k):
        j = _two_ranks({"L2_TP_WAIT_S": "10"}, tmp_path, 29615 + 20 * k)
        assert j["config"]["parallelism"] == "tp2" and j["scaling"] == "strong", j.get("note")
        assert "RCCL + peer-to-peer exchange:" in j["note"] and "RCCL collectives only:" in j["note"] and "no RCCL" in j["note"]
        # every formation is proved before it is timed: all ranks decoded the same tokens, and they are the real reference's
        pr = j["tp"]["proved_before_timing"]
        assert pr["same_on_every_rank"] and pr["equals_reference_golden"] is True and len(pr["tokens"]) == 16
    _retry_once(attempt)


def test_bench_falls_back_to_replicas_when_the_group_cannot_form(tmp_path):
    """No way to form the group (the meeting directory does not exist): the job measures replicas and says why."""
    j = _two_ranks({"L2_TP_IPC_DIR": str(tmp_path / "missing" / "dir")}, tmp_path, 29617)
    assert j["config"]["parallelism"] == "replicas2" and j["scaling"] == "weak" and "independent replicas instead" in j["note"]


def test_bench_gpus_flag_starts_its_own_ranks(tmp_path):
    """What the driver runs is plain `python bench.py --gpus N ...` -- no launcher around it.  The script, which has not
    touched the GPU, starts N ranks as a child `torch.distributed.run`, relays the one JSON line and the exit code.  Here
    N = 2 on the one GPU (L2_BENCH_FORCE_DEVICE=0), so the ranks end up meeting through files; the line must say 2 GPUs,
    tp2, name the tensor-parallel step (l2_tp_mode of every rank) and carry the note (llama2.ts:270, 292 are the reduce points)."""
    def attempt(k):
        env = dict(os.environ, L2_BENCH_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", L2_TP_WAIT_S="10")
        env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "llama2_7b_L2", "--steps", "16",
                            "--warmup", "2", "--no-cpu-baseline"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
        assert len(lines) == 1, lines
        j = json.loads(lines[0])
        assert j["n_gpus"] == 2 and j["config"]["parallelism"] == "tp2" and j["scaling"] == "strong" and j["value"] > 0, j.get("note")
        assert j["config"]["weights_mib"]["repacked"] > 0     # 7B width: the streaming kernels read their matrices in consumption order
        assert "note" in j and "no RCCL" in j["note"]
        assert j["tp"]["ranks"] == 2 and j["tp"]["sharded"] and j["tp"]["l2_tp_mode"] == [3] and j["tp"]["devices"] == [0, 0]
        # the stages that failed were given up and the next one started from FRESH worker processes; the line carries the rank's shard roofline
        f = j["tp"]["formation"]
        assert [x["stage"] for x in f] == ["p2p", "rccl", "file"] and [x["ok"] for x in f] == [False, False, True]
        rf = j["roofline"]
        assert rf["bound"] == "hbm" and rf["bytes_per_launch"] == 4 * (2 * (11008 // 2) * 4096 + 2 * 4096 + 11008 // 2) and 0 < rf["frac"] < 1
        assert "cpu_baseline" in j
    _retry_once(attempt)


def test_bench_gpus_one_stays_in_process():
    """--gpus 1 never spawns: same line as no flag at all (checked on the cheap shape)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "stories15M", "--steps", "16", "--warmup", "2",
                        "--no-cpu-baseline", "--no-pmc", "--no-extra"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    j = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 1 and j["config"]["parallelism"] == "single" and "tp" not in j
    assert j["config"]["weights_mib"]["repacked"] == 0


@pytest.mark.skipif(os.environ.get("L2_TEST_WIDE_PROCESS_GROUPS", "0") != "1",
                    reason="eight processes whose kernels wait for each other on ONE GPU: box-dependent (profiles/r05/tp_process_group_one_gpu_flakiness.txt); "
                           "L2_TEST_WIDE_PROCESS_GROUPS=1 runs it -- the two-rank tests above take the same supervised path and must pass")
def test_bench_gpus_8_forms_an_eight_rank_group_by_itself(tmp_path):
    """What the driver's scaling run does for N = 8, on the one GPU of this box: plain `python bench.py --gpus 8 --config llama2_7b_L2`,
    eight processes through bench's OWN launcher, and WITHOUT the development gate (L2_TEST_HOOKS unset): RCCL refuses eight ranks on
    one device, so the ranks meet through files under the library's narrow switch (L2_TP_FILE_RENDEZVOUS, set by bench.py around that
    attempt only).  The line must show an 8-rank sharded group that was proved against the reference's golden before it was timed, and
    the timed run's tokens checked too (llama2.ts:270, 292 are the reduce points)."""
    def attempt(k):
        env = dict(os.environ, L2_BENCH_FORCE_DEVICE="0", L2_TP_WAIT_S="20")
        for v in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "L2_TEST_HOOKS", "HSA_ENABLE_IPC_MODE_LEGACY", "L2_TP_IPC_DIR"):
            env.pop(v, None)        # HSA_ENABLE_IPC_MODE_LEGACY: bench.py sets its own default
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--config", "llama2_7b_L2", "--steps", "16",
                            "--warmup", "2", "--no-cpu-baseline"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500, env=env)
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        j = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][0])
        assert j["n_gpus"] == 8 and j["config"]["parallelism"] == "tp8" and j["scaling"] == "strong" and j["value"] > 0, j.get("note")
        assert j["tp"]["ranks"] == 8 and j["tp"]["sharded"] and j["tp"]["l2_tp_mode"] == [3] and j["tp"]["devices"] == [0] * 8
        pr = j["tp"]["proved_before_timing"]
        assert pr["same_on_every_rank"] and pr["equals_reference_golden"] is True
        assert j["parity"]["equal_to_reference_golden"] is True and j["parity"]["steps_checked"] == 16
        assert "tp_predicted" in j
    _retry_once(attempt)
