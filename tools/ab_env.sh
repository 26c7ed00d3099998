# A/B helper for the GPU box:  source tools/ab_env.sh; CFG=llama2_7b; run L2_TUNE_ROT=0; run L2_TUNE_ROT=5 L2_TUNE_GRIDCAP=768
# prints tok/s, ms per step and the end-to-end HBM fraction of one bench.py run under the given environment
export L2_TEST_HOOKS=1   # the development switches below only exist behind this gate
run() { echo "$* : $(env "$@" python bench.py --config $CFG --no-cpu-baseline --no-dropin --no-extra --no-pmc 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['hbm_frac_end_to_end'])")"; }
