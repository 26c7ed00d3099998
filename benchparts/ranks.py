"""`bench.py --gpus N`: how the N ranks are started and SUPERVISED so that the one JSON line always appears.

The tensor-parallel group (Llama-2-7B: heads / FFN rows sharded, the all-reduce points are llama2.ts:270 and :292) has never run on more
than one GPU in development, and neither RCCL nor a peer-to-peer wait across GPUs is guaranteed to FAIL when something is wrong -- an
ncclCommInitRank that hangs has no timeout.  So no process that supervises ever touches the GPU, and no process that touched the GPU is
ever reused for a second attempt:

  python bench.py --gpus N          (no launcher)   -> spawn_ranks(): starts `python -m torch.distributed.run ... bench.py` as a CHILD with
                                                        an overall deadline, relays the line and the exit code
  every rank the launcher starts    (WORLD_SIZE set) -> supervise(): a CPU-only supervisor (gloo among the supervisors).  For each way of
                                                        forming the group, in turn -- RCCL + peer-to-peer exchange, RCCL collectives only,
                                                        ranks meeting through files (no RCCL), independent replicas -- it starts ONE FRESH
                                                        worker process (`bench.py --worker-stage <s>`: benchparts/worker.py) that does the
                                                        GPU work, and watches the marks the worker prints: every phase (start, create,
                                                        prove, run) has a deadline.  A worker that fails, or sits out a deadline, is killed
                                                        with its whole process group; every supervisor hears about it (a key in the
                                                        launcher's store) and kills its own worker; the next stage starts from fresh
                                                        processes.  Rank 0's supervisor prints the worker's line with the notes of the
                                                        stages that failed -- or, if every stage failed, a line that says so (value null,
                                                        exit code 1).

Deadlines: L2_BENCH_STAGE_DEADLINES="start,create,prove,run" seconds (default 300,120,180,900).  Test hook (L2_TEST_HOOKS=1 only):
L2_BENCH_WORKER_STUB=<script> is started instead of the worker (tests/test_bench_cpu.py: a rank that sleeps forever)."""
import json
import os
import queue
import signal
import socket
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
PHASES = ("start", "create", "prove", "run")      # a worker's marks: "@@l2 started" ends start, "@@l2 created" ends create, "@@l2 proved" ends prove, its exit ends run
MARK = "@@l2 "


def deadlines():
    d = [300.0, 120.0, 180.0, 900.0]      # (start: a fresh box pages the image in under the first `import torch` -- minutes, not seconds)
    s = os.environ.get("L2_BENCH_STAGE_DEADLINES", "")
    if s:
        try:
            v = [float(x) for x in s.split(",")]
            if len(v) == 4 and all(x > 0 for x in v):
                d = v
        except ValueError:
            pass
    return dict(zip(PHASES, d))


def stage_plan(shards, ipc_base):
    """The ways of forming the group, in the order they are tried: (name, label of the note, environment of the workers)."""
    if not shards:
        return [("replicas", "independent replicas", {})]
    plan = [("p2p", "RCCL + peer-to-peer exchange", {}),
            ("rccl", "RCCL collectives only", {"L2_TP_ALLREDUCE": "rccl"})]
    if not ipc_base:     # (a meeting directory given from outside -- the one-GPU tests -- already makes the first stages meet through files)
        plan.append(("file", "file rendezvous + peer-to-peer exchange", {"L2_TP_FILE_RENDEZVOUS": "1"}))
    plan.append(("replicas", "independent replicas", {}))
    return plan


def free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def kill_group(proc):
    """The worker and everything it started (it leads a session of its own)."""
    if proc is None:
        return
    try:
        os.killpg(proc.pid, signal.SIGKILL)
    except (ProcessLookupError, PermissionError):
        pass
    try:
        proc.wait(timeout=10)
    except Exception:      # noqa: BLE001
        pass


class Watched:
    """A worker process and the marks / result line it prints."""

    def __init__(self, cmd, env):
        self.proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, start_new_session=True)
        self.lines = queue.Queue()
        self.reader = threading.Thread(target=self._read, daemon=True)
        self.reader.start()
        self.phase = 0
        self.since = time.monotonic()
        self.line = None
        self.why = ""

    def _read(self):
        for raw in self.proc.stdout:
            self.lines.put(raw.decode("utf8", "replace").rstrip("\n"))
        self.lines.put(None)

    def drain(self):
        """Take what the worker has printed so far; True once its stdout is closed."""
        while True:
            try:
                ln = self.lines.get_nowait()
            except queue.Empty:
                return False
            if ln is None:
                return True
            if ln.startswith(MARK):
                what = ln[len(MARK):].split(" ", 1)
                if what[0] in ("started", "created", "proved"):
                    self.phase = ("started", "created", "proved").index(what[0]) + 1
                    self.since = time.monotonic()
                elif what[0] == "failed":
                    self.why = what[1] if len(what) > 1 else "failed"
            elif ln.startswith("{") and '"metric"' in ln:
                self.line = ln
            elif ln.strip():
                print(ln, file=sys.stderr)      # anything else a worker printed is not the result line


def run_stage(cmd, env, dl, aborted, poll=0.1):
    """One worker for one stage: {"ok", "why", "line", "rc", "timed_out", "seconds"}.  `aborted()` -> a reason when another rank's stage failed."""
    t0 = time.monotonic()
    try:
        w = Watched(cmd, env)
    except OSError as e:
        return {"ok": False, "why": "could not start the worker: %s" % e, "line": None, "rc": None, "timed_out": False, "seconds": 0.0}
    out = {"ok": False, "why": "", "line": None, "rc": None, "timed_out": False}
    while True:
        closed = w.drain()
        rc = w.proc.poll()
        if rc is not None and closed:
            out["rc"] = rc
            out["line"] = w.line
            out["ok"] = rc in (0, 3)                     # (3: the line is there, the timed tokens are not the reference's -- bench.py's own verdict)
            if not out["ok"]:
                out["why"] = w.why or "the worker exited with code %d" % rc
            break
        phase = PHASES[min(w.phase, 3)]
        if time.monotonic() - w.since > dl[phase]:
            out["timed_out"] = True
            out["why"] = "no progress within %.0f s in phase '%s' (a hang, not a failure): the worker was killed" % (dl[phase], phase)
            break
        other = aborted()
        if other:
            out["why"] = "stopped: %s" % other
            break
        time.sleep(poll)
    kill_group(w.proc)
    out["seconds"] = round(time.monotonic() - t0, 1)
    return out


def failure_line(args, world, notes, formation):
    """Every way of running the ranks failed: the line says so (nothing was measured: value null)."""
    return json.dumps({"metric": "decode tokens/sec (whole job)", "value": None, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                       "ms_per_step": None, "higher_is_better": True, "scaling": None, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                       "config": {"workload": "%s batch-1 greedy decode" % args.config, "parallelism": None},
                       "failed": True, "error": "no stage of the multi-rank run produced a measurement", "note": "; ".join(notes), "tp": {"ranks": world, "formation": formation}})


def supervise(args, argv):
    """This process is one rank of the launcher (torch.distributed.run).  It stays a CPU-only supervisor.  Whatever goes wrong in the
    supervision itself (a rendezvous that fails, a store that dies) still ends with a line on rank 0 and a non-zero exit code."""
    try:
        return _supervise(args, argv)
    except Exception as e:      # noqa: BLE001 -- the line must appear
        import traceback
        traceback.print_exc()
        if int(os.environ.get("RANK", "0")) == 0:
            print(failure_line(args, int(os.environ.get("WORLD_SIZE", "1")), ["the supervisor of rank 0 failed: %s: %s" % (type(e).__name__, e)], []), flush=True)
        return 1


def _supervise(args, argv):
    import datetime
    import torch.distributed as dist
    dl = deadlines()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=sum(dl.values()) + 120))
    store = dist.distributed_c10d._get_default_store()
    shards = args.config.startswith("llama2_7b")
    ipc_base = os.environ.get("L2_TP_IPC_DIR")
    stub = os.environ.get("L2_BENCH_WORKER_STUB") if os.environ.get("L2_TEST_HOOKS") == "1" else None
    notes, formation, rc, printed = [], [], 1, False
    for idx, (name, label, stage_env) in enumerate(stage_plan(shards, ipc_base)):
        share = [None]
        if rank == 0:
            share[0] = {"port": free_port(), "meet": tempfile.mkdtemp(prefix="l2_meet_") if name == "file" else None}
        dist.broadcast_object_list(share, 0)
        env = dict(os.environ, L2_BENCH_WORKER="1", L2_BENCH_WORKER_PORT=str(share[0]["port"]), **stage_env)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if name == "file":
            env["L2_TP_IPC_DIR"] = share[0]["meet"]
        elif ipc_base and os.path.isdir(ipc_base) and name != "replicas":
            sub = os.path.join(ipc_base, "attempt%d" % (idx + 1))      # files of an earlier, failed formation must not be read again
            os.makedirs(sub, exist_ok=True)
            env["L2_TP_IPC_DIR"] = sub
        cmd = [sys.executable, stub or BENCH] + list(argv) + ["--worker-stage", name]
        key = "l2_bench_abort_%d" % idx

        def aborted():
            try:
                return store.get(key).decode("utf8", "replace") if store.check([key]) else ""
            except Exception:      # noqa: BLE001 -- a store hiccup must not end the supervision
                return ""

        mine = run_stage(cmd, env, dl, aborted)
        if not mine["ok"] and not mine["why"].startswith("stopped:"):
            try:
                store.set(key, "rank %d: %s" % (rank, mine["why"]))
            except Exception:      # noqa: BLE001
                pass
        every = [None] * world
        dist.all_gather_object(every, {"rank": rank, "ok": mine["ok"], "why": mine["why"], "rc": mine["rc"], "timed_out": mine["timed_out"], "seconds": mine["seconds"]})
        ok = all(r["ok"] for r in every)
        first = next((r for r in every if not r["ok"] and not r["why"].startswith("stopped:")), None) or next((r for r in every if not r["ok"]), None)
        formation.append({"stage": name, "ok": ok, "seconds": max(r["seconds"] for r in every),
                          "timed_out_ranks": [r["rank"] for r in every if r["timed_out"]], "why": None if ok else "rank %d: %s" % (first["rank"], first["why"])})
        if ok:
            rc = max(r["rc"] or 0 for r in every)
            if rank == 0:
                line = mine["line"]
                if line is None:
                    notes.append("%s: every rank finished but rank 0 printed no result line" % label)
                    ok = False
                    formation[-1]["ok"], formation[-1]["why"] = False, "rank 0 printed no result line"
                else:
                    j = json.loads(line)
                    if name == "file":
                        notes.append("the ranks met through files and exchange peer to peer (no RCCL)")
                    if name == "replicas" and shards:
                        notes.append("measured %d independent replicas instead" % world)
                    own = j.get("note")
                    allnotes = notes + ([own] if own else [])
                    if allnotes:
                        j["note"] = "; ".join(allnotes)
                    j.setdefault("tp", {})["formation"] = formation
                    print(json.dumps(j), flush=True)
                    printed = True
            flag = [ok]
            dist.broadcast_object_list(flag, 0)
            if flag[0]:
                break
            rc = 1
            continue
        notes.append("%s: rank %d: %s" % (label, first["rank"], first["why"]))
    else:
        rc = 1
    if rank == 0 and not printed:
        print(failure_line(args, world, notes, formation), flush=True)
        print("bench.py: every stage of the %d-rank run failed: %s" % (world, "; ".join(notes)), file=sys.stderr)
        rc = rc or 1
    try:
        dist.barrier()
        dist.destroy_process_group()
    except Exception:      # noqa: BLE001
        pass
    return rc


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: THIS process has not touched the GPU and never will; it starts
    `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a CHILD (never an exec), relays its one JSON line and
    its exit code -- and bounds it: the ranks supervise themselves (supervise()), so the bound only ends a launcher that itself hangs."""
    from .common import clean_child_env
    n = args.gpus
    port = os.environ.get("MASTER_PORT") or str(free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", port, BENCH] + list(argv)
    env = clean_child_env(HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    dl = deadlines()
    shards = args.config.startswith("llama2_7b")
    bound = float(os.environ.get("L2_BENCH_TOTAL_DEADLINE_S", 0)) or (len(stage_plan(shards, env.get("L2_TP_IPC_DIR"))) * sum(dl.values()) + 300.0)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=bound)
        rc = proc.returncode
    except subprocess.TimeoutExpired:
        try:
            proc.terminate()                      # the launcher forwards SIGTERM to its ranks (their workers die with them: PR_SET_PDEATHSIG) ...
            proc.wait(timeout=20)
        except Exception:      # noqa: BLE001
            pass
        kill_group(proc)                          # ... and whatever is left of its process group goes the hard way
        out, _ = proc.communicate()
        rc = 1
        print("bench.py: the launcher of the %d ranks did not end within %.0f s and was killed" % (n, bound), file=sys.stderr)
    line = None
    for ln in (out or b"").decode("utf8", "replace").splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        elif ln.strip():
            print(ln, file=sys.stderr)      # anything else a rank printed is not the result line
    if line is None:
        line = failure_line(args, n, ["the launcher ended (rc %s) without a result line" % rc], [])
        rc = rc or 1
    print(line)
    return rc
