// Does gfx950 protect the sources of a packed fp32 op against an overwrite by the NEXT instruction (write-after-read)?
// Sequence under test (seen in k_nrb_bwd_fused, where one accumulator came out nondeterministic):
//     v_pk_fma_f32 acc, a, b, acc  op_sel:[0,1,0] op_sel_hi:[1,0,1]
//     v_pk_mov_b32 a, b, c  op_sel:[1,0]          <- overwrites a, a source of the instruction before
// Many waves per SIMD so that issue timing varies.  hipcc --offload-arch=gfx950 -O3 -x hip pk_war_probe.cpp -o pk_war_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x2 acc = {0.f, 0.f}, a, b, c;
    const float t = threadIdx.x * 0.001f;
    for (int it = 0; it < (MODE == 3 ? 0 : iters); ++it) {
        a[0] = 1.0f + t; a[1] = 2.0f + t; b[0] = 3.0f; b[1] = 5.0f; c[0] = 7.0f; c[1] = 11.0f;
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
        if (MODE == 0) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]\n"
                         "v_pk_mov_b32 %1, %2, %3 op_sel:[1,0]\n" : "+v"(acc), "+v"(a) : "v"(b), "v"(c));
        } else if (MODE == 1) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]\n"
                         "s_nop 1\n"
                         "v_pk_mov_b32 %1, %2, %3 op_sel:[1,0]\n" : "+v"(acc), "+v"(a) : "v"(b), "v"(c));
        } else {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0\n"
                         "v_pk_mov_b32 %1, %3, %3\n" : "+v"(acc), "+v"(a) : "v"(b), "v"(c));
        }
        asm volatile("" : "+v"(a));
    }
    if (MODE == 3) {
        // the kernel's own shape: eight back-to-back packed multiply-adds reading a, then packed moves overwriting a
        f32x2 c2[8];
        for (int i = 0; i < 8; ++i) c2[i] = f32x2{0.f, 0.f};
        f32x2 d = {13.f, 17.f};
        for (int it = 0; it < iters; ++it) {
            a[0] = 1.0f + t; a[1] = 2.0f + t; b[0] = 3.0f; b[1] = 5.0f; c[0] = 7.0f; c[1] = 11.0f;
            asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            asm volatile("v_pk_fma_f32 %0, %8, %9, %0\n"
                         "v_pk_fma_f32 %1, %8, %10, %1 op_sel_hi:[0,1,1]\n"
                         "v_pk_fma_f32 %2, %8, %10, %2 op_sel:[1,0,0]\n"
                         "v_pk_fma_f32 %3, %8, %11, %3 op_sel_hi:[0,1,1]\n"
                         "v_pk_fma_f32 %4, %8, %11, %4 op_sel:[1,0,0]\n"
                         "v_pk_fma_f32 %5, %8, %10, %5 op_sel_hi:[0,1,1]\n"
                         "v_pk_fma_f32 %6, %8, %11, %6 op_sel:[1,0,0]\n"
                         "v_pk_fma_f32 %7, %8, %9, %7 op_sel:[0,1,0] op_sel_hi:[1,0,1]\n"
                         "v_pk_mov_b32 %8, %9, %10 op_sel:[1,0]\n"
                         "v_pk_mov_b32 %10, %10, %11 op_sel:[1,0]\n"
                         : "+v"(c2[0]), "+v"(c2[1]), "+v"(c2[2]), "+v"(c2[3]), "+v"(c2[4]), "+v"(c2[5]), "+v"(c2[6]), "+v"(c2[7]), "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            asm volatile("" : "+v"(a), "+v"(c));
        }
        acc = c2[7];
    }
    out[(blockIdx.x * 256 + threadIdx.x) * 2] = acc[0];
    out[(blockIdx.x * 256 + threadIdx.x) * 2 + 1] = acc[1];
}

template <int MODE> void run(const char* name) {
    const int grid = 256 * 8, iters = 4096, n = grid * 256;
    float* out; (void)hipMalloc(&out, n * 2 * 4);
    k<MODE><<<grid, 256>>>(out, iters);
    float* h = new float[n * 2];
    (void)hipMemcpy(h, out, n * 2 * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int i = 0; i < n; ++i) {
        const float t = (i % 256) * 0.001f;
        // MODE 0/1: lo = a.lo * b.hi, hi = a.hi * b.lo;  MODE 2: lo = a.lo * b.lo, hi = a.hi * b.hi -- accumulated `iters` times in fp32
        float e0 = 0.f, e1 = 0.f;
        const float p0 = MODE == 2 ? (1.0f + t) * 3.0f : (1.0f + t) * 5.0f, p1 = MODE == 2 ? (2.0f + t) * 5.0f : (2.0f + t) * 3.0f;
        for (int it = 0; it < iters; ++it) { e0 = fmaf(1.0f + t, MODE == 2 ? 3.0f : 5.0f, e0); e1 = fmaf(2.0f + t, MODE == 2 ? 5.0f : 3.0f, e1); }
        (void)p0; (void)p1;
        if (h[2 * i] != e0 || h[2 * i + 1] != e1) { if (bad < 3) printf("   lane %d: got %.3f %.3f want %.3f %.3f\n", i, h[2 * i], h[2 * i + 1], e0, e1); ++bad; }
    }
    printf("%-46s %ld of %d lanes wrong\n", name, bad, n);
    (void)hipFree(out); delete[] h;
}
int main() {
    run<0>("pk_fma (op_sel) then pk_mov over its source");
    run<1>("the same with s_nop 1 between");
    run<2>("pk_fma then v_mov over its source (lo half)");
    run<3>("eight pk_fma in a row, then pk_mov over the source");
    return 0;
}
