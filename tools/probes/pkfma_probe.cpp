// Issue rate of v_pk_fma_f32 against v_fma_f32 on gfx950: one wave per SIMD slot (256 threads x 4 waves/SIMD), long unrolled
// chains of INDEPENDENT accumulators, wall-clock by HIP events.  hipcc --offload-arch=gfx950 -O3 -x hip pkfma_probe.cpp -o pkfma_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f32x2 acc[8];
    float s[16];
    for (int i = 0; i < 8; ++i) { acc[i][0] = threadIdx.x * 0.001f + i; acc[i][1] = i; }
    for (int i = 0; i < 16; ++i) s[i] = threadIdx.x * 0.002f + i;
    f32x2 w; w[0] = a; w[1] = b;
    f32x2 x; x[0] = b; x[1] = a;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(w), "v"(x));
            } else if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i]) : "v"(a), "v"(b));
            } else if (MODE == 2) {   // op_sel broadcast form
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[i]) : "v"(w), "v"(x));
            } else {                  // v_pk_mul_f32
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(acc[i]) : "v"(w));
            }
        }
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1];
    for (int i = 0; i < 16; ++i) r += s[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE> void run(const char* name, int flops_per_inst, int insts_per_round) {
    float* out; hipMalloc(&out, 256 * 4096 * 4);
    const int iters = 2000, grid = 256 * 4;      // 4 WGs of 4 waves per CU: 4 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, 256>>>(out, 10, 1.0001f, 0.5f);
    hipEventRecord(e0);
    k<MODE><<<grid, 256>>>(out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winst = (double)grid * 4 * iters * 8 * insts_per_round;     // wave-instructions
    // cycles per wave-instruction per SIMD at 2.4 GHz: 1024 SIMDs
    printf("%-28s %.3f ms  %.2f cycles per wave-instruction per SIMD (2.4 GHz), %.1f TFLOP/s\n", name, ms,
           ms * 1e-3 * 2.4e9 / (winst / 1024), winst * 64 * flops_per_inst / (ms * 1e-3) / 1e12);
    hipFree(out);
}
int main() {
    run<1>("v_fma_f32", 2, 16);
    run<0>("v_pk_fma_f32", 4, 8);
    run<2>("v_pk_fma_f32 op_sel bcast", 4, 8);
    run<3>("v_pk_mul_f32", 2, 8);
    return 0;
}
