#!/bin/bash
# AMD_DIRECT_DISPATCH=0 (runtime submits from its own thread) against the default, on the real decode: same box, alternating
source tools/ab_env.sh
for r in 1 2; do
for CFG in stories110M stories15M; do
  run X=0
  run AMD_DIRECT_DISPATCH=0
done
done
CFG=llama2_7b
run X=0
run AMD_DIRECT_DISPATCH=0
python - <<'PY'
import os, subprocess, sys, json
for env in ({}, {"AMD_DIRECT_DISPATCH": "0"}):
    e = dict(os.environ); e.update(env)
    out = subprocess.run([sys.executable, "bench.py", "--config", "stories110M", "--no-cpu-baseline", "--no-extra", "--no-pmc"], env=e, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    j = json.loads(out)
    print(env, "value", j["value"], "dropin", j.get("dropin_tok_s"))
PY
