// Narrow-level (C = 4, 8) ResidualConv2dBlock on the vector ALUs: see conv_small.hip.
#pragma once
#include <hip/hip_runtime.h>

int tt_small_rb_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* y, float* h1,
                    int B, int C, int H, int T, int dilation, hipStream_t st);
// writes dA1 to ws and dx; accumulates db1, dw2, db2; the caller runs the MFMA weight gradient for dw1
int tt_small_rb_bwd(const float* x, const float* h1, const float* dy, const float* w1, const float* b1, const float* w2,
                    const float* b2, float* dx, float* dw1, float* db1, float* dw2, float* db2, float* ws, float* scratch, int B, int C, int H,
                    int T, int dilation, hipStream_t st);
