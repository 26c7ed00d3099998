cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle_lib as O, synth_tokenizer
from llama2_ts_amd import configs
os.makedirs("/tmp/cli110", exist_ok=True)
O.synth_write(configs.header("stories110M"), 1, "/tmp/cli110/model.bin")
synth_tokenizer.write("/tmp/cli110/tokenizer.bin")
PY
cd /tmp/cli110
H=$GRAFT_REPO_ROOT/llama2.ts_amd/host/llama2.mjs
time node $H model.bin -t 0 -s 1 -n 40 | tail -3
time L2_NATIVE_LOADER=1 L2_DEVICE_GREEDY=1 node $H model.bin -t 0 -s 1 -n 256 | tail -2
