run() { echo "$* : $(env "$@" python bench.py --config $CFG --no-cpu-baseline --no-dropin --no-extra 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['hbm_frac_end_to_end'])")"; }
