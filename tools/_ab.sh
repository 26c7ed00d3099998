source tools/ab_env.sh
CFG=llama2_7b
for i in 1 2; do
run L2_TUNE_NWAVES=0
run L2_TUNE_NWAVES=8
done
python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "7b_L2 or repacked" 2>&1 | tail -2
L2_TUNE_NWAVES=8 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "7b_L2 or repacked" 2>&1 | tail -2
