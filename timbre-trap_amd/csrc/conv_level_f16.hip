// fp16 build of conv_level_bf16.hip: the same kernels with e16 = _Float16 (csrc/bf16_common.h), entry points with the suffix _h
// (csrc/e16_names.h, include/ttrap.h "fp16 twins").  The reference's own train step runs in this dtype (experiments/train.py:415).
#define TT_F16 1
#include "conv_level_bf16.hip"
