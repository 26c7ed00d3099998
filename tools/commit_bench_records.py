"""After a single-GPU `python bench.py` whose line is kept under profiles/: write the two records a MULTI-rank run quotes instead of measuring
(every GPU is busy being a rank, every host core a supervisor) -- profiles/tp_predicted.json (a rank's shard step alone, benchparts/single.py:
tp_prediction) and profiles/cpu_baseline_n1.json (cpu_baseline of the N = 1 run, per configuration; benchparts/worker.py copies it).

  python tools/commit_bench_records.py profiles/r06/bench_default_llama2_7b.json [more lines ...]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(paths):
    cpu_path = os.path.join(ROOT, "profiles", "cpu_baseline_n1.json")
    tp_path = os.path.join(ROOT, "profiles", "tp_predicted.json")
    cpu = json.load(open(cpu_path)) if os.path.exists(cpu_path) else {}
    tp = json.load(open(tp_path)) if os.path.exists(tp_path) else {}
    for path in paths:
        line = [ln for ln in open(path).read().splitlines() if ln.startswith("{") and '"metric"' in ln][-1]
        j = json.loads(line)
        if "parsed" in j:      # a driver record (BENCH_rNN.json)
            j = j["parsed"]
        name = j["config"]["workload"].split(" ")[0]
        rel = os.path.relpath(os.path.abspath(path), ROOT)
        for nm, blk in ((name, j), ("stories110M", j.get("stories110M") or {})):
            cb = blk.get("cpu_baseline")
            if cb and cb.get("value"):
                cpu[nm] = dict({k: v for k, v in cb.items() if k != "js_port"}, js_port_tok_s=(cb.get("js_port") or {}).get("value"), run="python bench.py, n_gpus 1 (%s)" % rel)
        if j.get("n_gpus") == 1 and "tp_predicted" in j and "2" in j["tp_predicted"]:
            tp[name] = j["tp_predicted"]
            tp["measured_by"] = "python bench.py (single MI355X, %s)" % rel
    json.dump(cpu, open(cpu_path, "w"), indent=1)
    json.dump(tp, open(tp_path, "w"), indent=1)
    print("cpu_baseline_n1:", {k: v["value"] for k, v in cpu.items()}, " tp_predicted:", [k for k in tp if k != "measured_by"])


if __name__ == "__main__":
    main(sys.argv[1:])
