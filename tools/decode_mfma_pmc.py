"""Counter evidence for "the decode path issues no MFMA" (llama2.ts:196-203 is a batch-1 GEMV: no second dimension for
a matrix core to contract over).  Separate `rocprofv3 --kernel-trace --pmc <counter>` passes -- one counter per pass, the program directly
behind `--` -- over `python3 bench.py --trace-child --config <name>` (24 greedy tokens, eager launches: the library's own queue stands
down under a profiler's tool library), reduced to one row per kernel of the step: launches, mean of every counter per launch.

  python3 tools/decode_mfma_pmc.py <config> <out.json>        (on the GPU box; `cd /tmp && export TMPDIR=/tmp` first)
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COUNTERS = ["SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_BUSY_CYCLES", "SQ_WAVES"]


def one_pass(name, ctr, work):
    d = os.path.join(work, ctr)
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "p", "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--trace-child", "--config", name]
    env = dict(os.environ, TMPDIR="/tmp", L2_USE_GRAPH="0", L2_PROFILE_SYNC="1", L2_TEST_HOOKS="1")
    r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=900)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if r.returncode != 0 or not files:
        return None, "rc %d: %s" % (r.returncode, r.stderr.decode("utf8", "replace")[-300:])
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(files[0])):
        if row["Counter_Name"] == ctr:
            acc[row["Kernel_Name"].split("(")[0].replace("void l2k::", "").replace("l2k::", "")].append(float(row["Counter_Value"]))
    return acc, ""


def main(name, out_path):
    work = tempfile.mkdtemp(prefix="l2_mfma_", dir="/tmp")
    kernels, notes = collections.defaultdict(dict), {}
    try:
        for ctr in COUNTERS:
            acc, why = one_pass(name, ctr, work)
            if acc is None:
                notes[ctr] = why
                continue
            for k, vals in acc.items():
                kernels[k]["launches"] = len(vals)
                kernels[k][ctr] = sum(vals) / len(vals)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    step = {k: v for k, v in kernels.items() if "synth" not in k and "pack_kernel" not in k}
    total_mfma = sum(v.get("SQ_INSTS_MFMA", 0.0) * v["launches"] for v in step.values())
    total_mops = sum(v.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) * v["launches"] for v in step.values())
    total_valu = sum(v.get("SQ_INSTS_VALU", 0.0) * v["launches"] for v in step.values())
    rec = {"what": "rocprofv3 --kernel-trace --pmc <one counter per pass> -- python3 bench.py --trace-child --config %s (24 greedy tokens from BOS, eager launches)" % name,
           "config": name, "counters": COUNTERS, "failed_passes": notes,
           "decode_step_totals": {"SQ_INSTS_MFMA": total_mfma, "SQ_INSTS_VALU_MFMA_MOPS_F64": total_mops, "SQ_INSTS_VALU": total_valu},
           "mfma_insts": int(total_mfma), "kernels_per_launch_mean": step}
    json.dump(rec, open(out_path, "w"), indent=1)
    print("%s: %d kernels of the step, MFMA instructions %d, MFMA F64 MOPS %d, VALU instructions %.3g" % (name, len(step), total_mfma, total_mops, total_valu))
    for k, v in sorted(step.items()):
        print("  %-58s x%-5d MFMA %-6.0f MOPS_F64 %-6.0f VALU %-12.0f busy cycles %.0f" % (k[:58], v["launches"], v.get("SQ_INSTS_MFMA", -1), v.get("SQ_INSTS_VALU_MFMA_MOPS_F64", -1), v.get("SQ_INSTS_VALU", -1), v.get("SQ_BUSY_CYCLES", -1)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
