source tools/ab_env.sh
CFG=llama2_7b
python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "7b_L2 or repacked or random or golden" 2>&1 | tail -2
L2_TUNE_BALANCE=0 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "7b_L2 or repacked" 2>&1 | tail -2
for i in 1 2 3; do
run L2_TUNE_BALANCE=1
run L2_TUNE_BALANCE=0
done
