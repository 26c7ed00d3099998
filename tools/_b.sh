mkdir -p gpurun_out/r03
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03/bench_now.json
python - <<EOF
import json; j=json.load(open("gpurun_out/r03/bench_now.json"))
print(j["roofline"]); print({k:v for k,v in j.items() if k not in ("stories110M","roofline","config","cpu_baseline","stories15M")})
EOF
