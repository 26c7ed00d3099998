#!/bin/bash
# L2 carry from the attention launch (its free CUs pull the layer's wo, w1/w3, w2 into their XCDs' L2) against a plain attention launch; the fused
# QKV + attention launch on and off (the carry applies to the plain attention launch only): same box, alternating
source tools/ab_env.sh
for CFG in stories110M stories15M; do
  run L2_CARRY=0
  run L2_CARRY=1
  run L2_CARRY=0 L2_FUSE_QKV_ATTN=0
  run L2_CARRY=1 L2_FUSE_QKV_ATTN=0
done
