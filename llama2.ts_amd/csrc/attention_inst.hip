// attention_inst.hip -- compiles the attention family's kernel instances (attention_inst.hip.h) as a translation unit of its own.
#define L2_ATTN_INST
#define L2_NO_PLAIN_KERNELS      // (the plain kernels of the shared headers are defined in llama2_hip.hip)
#include "attention_inst.hip.h"
