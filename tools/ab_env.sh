# same-box A/B of one environment switch: bash tools/ab_env.sh L2_PREFETCH_EPI   (runs 0,0,1,0,1; the first run warms the box up)
v=$1
run() { echo "$* : $(env "$@" python bench.py --no-cpu-baseline --no-dropin 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['hbm_frac_end_to_end'], j['stories110M']['value'], j['long_context']['value'])")"; }
run $v=0; run $v=0; run $v=1; run $v=0; run $v=1
