// Narrow-level (C = 4, 8) ResidualConv2dBlock on the vector ALUs: see conv_small.hip.
#pragma once
#include <hip/hip_runtime.h>

int tt_small_rb_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* y, float* h1,
                    int B, int C, int H, int T, int dilation, hipStream_t st);
// Two ways back: TT_SMALL_BWD_DID_DW1 -- the fused kernel ran (dx written; dw1, db1, dw2, db2 accumulated; ws untouched);
// 0 -- the three-kernel path wrote dA1 to ws and dx and accumulated db1, dw2, db2: the caller still runs the MFMA weight
// gradient for dw1 from ws.  Anything else is an error code.  `scratch`: tt_wgrad_scratch_floats() floats.
#define TT_SMALL_BWD_DID_DW1 0x7fff0001
int tt_small_rb_bwd(const float* x, const float* h1, const float* dy, const float* w1, const float* b1, const float* w2,
                    const float* b2, float* dx, float* dw1, float* db1, float* dw2, float* db2, float* ws, float* scratch, int B, int C, int H,
                    int T, int dilation, hipStream_t st);
