"""Stand-in for `bench.py --worker-stage <s>` (tests/test_bench_cpu.py; started by benchparts/ranks.py when L2_BENCH_WORKER_STUB names it
and L2_TEST_HOOKS=1): a rank's worker without a GPU.  L2_STUB_HANG="stage:phase,..." -- sleep forever when that stage reaches that
phase (start / create / prove / run), like an ncclCommInitRank that never returns; L2_STUB_FAIL="stage:rank,..." -- that rank reports
`@@l2 failed` at create and exits; L2_STUB_PIDS=<file>: every stub appends its pid (the test checks that the hung ones are gone)."""
import json
import os
import sys
import time

stage = sys.argv[sys.argv.index("--worker-stage") + 1]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
hang = dict(x.split(":") for x in os.environ.get("L2_STUB_HANG", "").split(",") if x)
fail = [x.split(":") for x in os.environ.get("L2_STUB_FAIL", "").split(",") if x]
if os.environ.get("L2_STUB_PIDS"):
    with open(os.environ["L2_STUB_PIDS"], "a") as f:
        f.write("%d %s %d\n" % (os.getpid(), stage, rank))


def phase(name, mark):
    if hang.get(stage) == name:
        time.sleep(100000)
    if mark:
        print("@@l2 " + mark, flush=True)


assert os.environ.get("L2_BENCH_WORKER") == "1" and os.environ.get("L2_BENCH_WORKER_PORT")
phase("start", "started")
for st, r in fail:
    if st == stage and int(r) == rank:
        print("@@l2 failed L2Error: stub rank %d cannot form the group in stage %s" % (rank, stage), flush=True)
        sys.exit(4)
phase("create", "created")
phase("prove", "proved")
phase("run", None)
if rank == 0:
    print(json.dumps({"metric": "decode tokens/sec (whole job)", "value": 123.0, "unit": "tokens/s", "n_gpus": world, "steps": 4, "warmup": 1, "ms_per_step": 8.13,
                      "higher_is_better": True, "scaling": "weak" if stage == "replicas" else "strong", "vs_baseline": None, "dtype": "f64", "data": "stub",
                      "config": {"workload": "stub", "parallelism": ("replicas%d" if stage == "replicas" else "tp%d") % world},
                      "tp": {"ranks": world, "stage": stage, "sharded": stage != "replicas", "env": {k: os.environ.get(k) for k in ("L2_TP_ALLREDUCE", "L2_TP_FILE_RENDEZVOUS", "L2_TP_IPC_DIR")}},
                      "roofline": {"bound": "hbm"}, "cpu_baseline": {"value": 1.0}}), flush=True)
