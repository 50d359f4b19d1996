// Does v_mfma_f32_16x16x32_f16 honour SUBNORMAL fp16 inputs on gfx950, and does v_cvt_f16_f32 produce them?
// hipcc --offload-arch=gfx950 -O2 tools/diag/mfma_denorm.hip -o /tmp/mfma_denorm && /tmp/mfma_denorm
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, float tiny, float big) {
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)0.f; b[j] = (_Float16)0.f; }
    a[0] = (_Float16)tiny;                 // row (lane & 15), k = 8 (lane >> 4)
    b[0] = (_Float16)big;
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; out[2] = (float)((_Float16)(tiny * 0.001f)); }
}
int main() {
    float* d; hipMalloc(&d, 16);
    const float tiny = 9.5367431640625e-07f;   // 2^-20: subnormal in fp16 (exactly representable)
    k<<<1, 64>>>(d, tiny, 1024.f);
    float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
    // 4 k-groups each contribute tiny * big
    printf("mfma(2^-20 x 1024 x 4 k-groups) = %g (honoured: %g)   cvt_f16(2^-20) = %g   cvt_f16(2^-20 / 1000) = %g (quantum 5.96e-8)\n",
           h[0], 4 * tiny * 1024.f, h[1], h[2]);
    return 0;
}
