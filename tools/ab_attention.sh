# same-box A/B of the attention launch forms: bash tools/ab_attention.sh
run() { echo "$* : $(env "$@" python bench.py --no-cpu-baseline --no-dropin 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['hbm_frac_end_to_end'], j['stories110M']['value'])")"; }
run L2_ATTN_PRE=0
run L2_ATTN_PRE=1
run L2_ATTN_PRE=2
run L2_ATTN_PRE=3
run L2_ATTN_PRE=0
run L2_ATTN_PRE=1
