#!/bin/bash
# L2 carry (every latency-form GEMV launch touches what the next one will ask for first) against none: same box, alternating
source tools/ab_env.sh
for r in 1 2; do
for CFG in stories110M stories15M; do
  run L2_CARRY=0
  run L2_CARRY=1
done
done
