"""rocprofv3 legs of bench.py: HBM traffic of the dominant kernel (two --pmc passes) and its duration by kernel trace, each over a child of bench.py that is started BEFORE this process touches the GPU."""
import csv
import glob
import os
import shutil
import subprocess
import sys
import tempfile


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from llama2_ts_amd import configs, runtime  # noqa: E402

from .common import clean_child_env, under_profiler  # noqa: E402


def pmc_child(name, seed):
    """Target of the counter passes: the model of this config, one forward, a few launches of the dominant kernel."""
    ctx = runtime.Context(configs.header(name))
    ctx.synth_fill(seed)
    ctx.forward(1, 0)
    ctx.bench_gemv(runtime.T_W1, ctx.cfg.n_layers // 2, 6)
    ctx.close()


def pmc_traffic(name, seed):
    """FETCH_SIZE and WRITE_SIZE in SEPARATE passes (kernel trace only), corrected as the MI355X guide prescribes:
    both are in KiB and FETCH_SIZE reports exactly half of a 16-byte-per-lane coalesced stream on gfx950."""
    if under_profiler():
        return None, "skipped: this run is itself being profiled"
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not found"
    out = {}
    work = tempfile.mkdtemp(prefix="l2_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(work, ctr)
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "p", "--",
                   sys.executable, BENCH, "--pmc-child", "--config", name, "--seed", str(seed)]
            env = clean_child_env(TMPDIR="/tmp", L2_USE_GRAPH="0")
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=600)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, "rocprofv3 --pmc %s failed (rc %d)" % (ctr, r.returncode)
            vals = []
            for row in csv.DictReader(open(files[0])):
                kn = row["Kernel_Name"]
                if row["Counter_Name"] == ctr and ("phase_kernel<2," in kn or "phase_small_kernel<2," in kn):
                    vals.append(float(row["Counter_Value"]))
            if len(vals) < 3:
                return None, "no launches of the dominant kernel in the %s pass" % ctr
            vals = vals[2:]   # the first launches follow a forward: drop them like warm-up
            out[ctr] = sum(vals) / len(vals)
    except Exception as e:   # noqa: BLE001 -- a missing profiler must not fail the benchmark
        return None, "pmc pass: %r" % (e,)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    total = out["FETCH_SIZE"] * 1024.0 * 2.0 + out["WRITE_SIZE"] * 1024.0
    return int(total), ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of this run (FETCH_SIZE KiB x 1024 x 2 + "
                        "WRITE_SIZE KiB x 1024; per launch, mean of %d)" % len(vals))


def mfma_instructions(name, seed):
    """The decode path issues no MFMA (batch-1 GEMV: no second dimension for a matrix core, llama2.ts:196-203) -- as a COUNTER, not a
    sentence: one more `rocprofv3 --pmc SQ_INSTS_MFMA` pass over the same child (one whole forward + launches of the dominant kernel),
    summed over every kernel of the library it ran.  (profiles/r06/decode_mfma_pmc_*.json: the same per kernel over 24 decoded tokens,
    with SQ_INSTS_VALU_MFMA_MOPS_F64, SQ_VALU_MFMA_BUSY_CYCLES and SQ_INSTS_VALU beside it.)"""
    if under_profiler():
        return None, "skipped: this run is itself being profiled"
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not found"
    work = tempfile.mkdtemp(prefix="l2_mfma_", dir="/tmp")
    try:
        cmd = [exe, "--kernel-trace", "--pmc", "SQ_INSTS_MFMA", "--output-format", "csv", "-d", work, "-o", "p", "--",
               sys.executable, BENCH, "--pmc-child", "--config", name, "--seed", str(seed)]
        r = subprocess.run(cmd, cwd="/tmp", env=clean_child_env(TMPDIR="/tmp", L2_USE_GRAPH="0"), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=600)
        files = glob.glob(os.path.join(work, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            return None, "rocprofv3 --pmc SQ_INSTS_MFMA failed (rc %d)" % r.returncode
        total, launches = 0.0, 0
        for row in csv.DictReader(open(files[0])):
            if row["Counter_Name"] == "SQ_INSTS_MFMA" and "l2k::" in row["Kernel_Name"] and "synth_fill" not in row["Kernel_Name"] and "pack_kernel" not in row["Kernel_Name"]:
                total += float(row["Counter_Value"]); launches += 1
        if not launches:
            return None, "no kernel of the library in the SQ_INSTS_MFMA pass"
        return int(total), "rocprofv3 --pmc SQ_INSTS_MFMA over a child of this run: one whole forward + the dominant kernel, %d launches of the library's kernels" % launches
    except Exception as e:   # noqa: BLE001 -- a missing profiler must not fail the benchmark
        return None, "pmc pass: %r" % (e,)
    finally:
        shutil.rmtree(work, ignore_errors=True)


def kernel_trace_us(name, seed):
    """Average duration of the dominant kernel as a kernel trace reports it (rocprofv3 --kernel-trace, no counters): a child of
    this script decodes 24 tokens with eager launches.  Quoted next to the HIP-event figure: on a 5 us kernel the event pair
    itself costs about 1 us."""
    if under_profiler():
        return None
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    work = tempfile.mkdtemp(prefix="l2_kt_", dir="/tmp")
    try:
        cmd = [exe, "--kernel-trace", "--output-format", "csv", "-d", work, "-o", "k", "--",
               sys.executable, BENCH, "--trace-child", "--config", name, "--seed", str(seed)]
        env = clean_child_env(TMPDIR="/tmp", L2_USE_GRAPH="0", L2_PROFILE_SYNC="1", L2_TEST_HOOKS="1")
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
        files = glob.glob(os.path.join(work, "**", "*kernel_trace.csv"), recursive=True)
        if r.returncode != 0 or not files:
            return None
        durs = []
        for row in csv.DictReader(open(files[0])):
            kn = row["Kernel_Name"]
            if "phase_kernel<2," in kn or "phase_small_kernel<2," in kn:
                durs.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)
        durs = durs[len(durs) // 4:]     # the first quarter is warm-up (clocks, caches)
        return round(sum(durs) / len(durs), 3) if durs else None
    except Exception:   # noqa: BLE001 -- a missing profiler must not fail the benchmark
        return None
    finally:
        shutil.rmtree(work, ignore_errors=True)


def trace_child(name, seed):
    ctx = runtime.Context(configs.header(name))
    ctx.synth_fill(seed)
    ctx.decode_greedy(1, 0, min(24, configs.header(name)[6]))
    ctx.close()


