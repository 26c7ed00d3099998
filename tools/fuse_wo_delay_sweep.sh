for d in 0 1 2 4 6 8; do echo "delay $d: $(L2_FUSE_WO_DELAY=$d python bench.py --no-cpu-baseline --no-dropin --no-extra 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])")"; done
