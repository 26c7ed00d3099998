#!/bin/bash
# the chained FFN launch (wo -> w1/w3 -> w2 in one launch) against three launches: same box, alternating
source tools/ab_env.sh
for CFG in stories110M stories15M; do
  run L2_FFN_CHAIN=0
  run L2_FFN_CHAIN=1
  run L2_FFN_CHAIN=1 L2_CHAIN_NAP=1
  run L2_FFN_CHAIN=1 L2_CHAIN_NAP=2
  run L2_FFN_CHAIN=1 L2_CHAIN_NAP=3
done
