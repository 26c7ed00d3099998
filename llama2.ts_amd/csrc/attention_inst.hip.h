// attention_inst.hip.h -- the attention family's kernel instances, listed ONCE: llama2_hip.hip sees them as explicit instantiation
// DECLARATIONS (extern template: it launches them, it does not compile them), attention_inst.hip as explicit instantiation DEFINITIONS.
// The family is ~40 % of the library's device code; as a translation unit of its own it compiles beside the rest (make -j).
// Which instances exist is decided by launch.hip.h (launch_attn_tile, launch_qkv_attn_xv, launch_attn_wo): keep the two in step --
// an instance launched there and missing here fails at LINK time (undefined kernel stub), never silently.
#pragma once
#include "attention.hip.h"

#ifndef L2_ATTN_INST
#define L2_ATTN_INST extern      // declarations by default
#endif

namespace l2k {
#define L2_AT_INST(LR, NW, NT) \
  L2_ATTN_INST template __global__ void attn_tile_kernel<LR, NW, NT>(const AttnArgs); \
  L2_ATTN_INST template __global__ void pf_attn_tile_kernel<LR, NW, NT>(const AttnArgs, int);
L2_AT_INST(4, 4, 16) L2_AT_INST(8, 4, 16) L2_AT_INST(16, 4, 16) L2_AT_INST(32, 8, 8) L2_AT_INST(64, 4, 16)
#undef L2_AT_INST
L2_ATTN_INST template __global__ void qkv_attn_small_kernel<2, 16, 8>(const PhaseArgs, const AttnArgs, const int);
L2_ATTN_INST template __global__ void qkv_attn_small_kernel<3, 16, 8>(const PhaseArgs, const AttnArgs, const int);
L2_ATTN_INST template __global__ void qkv_attn_small_kernel<4, 16, 8>(const PhaseArgs, const AttnArgs, const int);
L2_ATTN_INST template __global__ void attn_wo_kernel<2, 2, 32, 8, 8>(const AttnArgs, const PhaseArgs, const int);
L2_ATTN_INST template __global__ void attn_wo_kernel<4, 2, 32, 8, 8>(const AttnArgs, const PhaseArgs, const int);
L2_ATTN_INST template __global__ void attn_wo_kernel<8, 2, 32, 8, 8>(const AttnArgs, const PhaseArgs, const int);
}  // namespace l2k
