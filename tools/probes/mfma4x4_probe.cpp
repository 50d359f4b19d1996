// Probe of v_mfma_f32_4x4x1_16b_f32 on gfx950: operand / result lane mapping and issue rate.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma4x4_probe tools/probes/mfma4x4_probe.cpp && /tmp/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_layout(float* out) {
    const int l = threadIdx.x;
    // A: block b = l / 4, row i = l % 4 -> value 1 + i + 10 b ;  B: block b, column j = l % 4 -> value 100 + j + 1000 b  (hypothesis)
    const float a = 1.f + (l & 3) + 10.f * (l >> 2), b = 100.f + (l & 3) + 1000.f * (l >> 2);
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}

__global__ void k_rate(float* out, int iters) {
    const int l = threadIdx.x;
    float a = 1.f + l, b = 0.5f;
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + l] = c0[0] + c1[1] + c2[2] + c3[3];
}

__global__ void k_rate_dep(float* out, int iters) {      // one dependent accumulator chain
    const int l = threadIdx.x;
    float a = 1.f + l, b = 0.5f;
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 4 * iters; ++i) c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
    out[blockIdx.x * blockDim.x + l] = c0[0];
}

__global__ void k_rate_fma(float* out, int iters) {      // the VALU it would replace: 4 fmas per k
    const int l = threadIdx.x;
    float x = 0.5f + l * 1e-3f, w0 = 1.1f, w1 = 1.2f, w2 = 1.3f, w3 = 1.4f;
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0, c0 = 0, c1 = 0, c2 = 0, c3 = 0, d0 = 0, d1 = 0, d2 = 0, d3 = 0;
    for (int i = 0; i < iters; ++i) {
        a0 = fmaf(x, w0, a0); a1 = fmaf(x, w1, a1); a2 = fmaf(x, w2, a2); a3 = fmaf(x, w3, a3);
        b0 = fmaf(x, w0, b0); b1 = fmaf(x, w1, b1); b2 = fmaf(x, w2, b2); b3 = fmaf(x, w3, b3);
        c0 = fmaf(x, w0, c0); c1 = fmaf(x, w1, c1); c2 = fmaf(x, w2, c2); c3 = fmaf(x, w3, c3);
        d0 = fmaf(x, w0, d0); d1 = fmaf(x, w1, d1); d2 = fmaf(x, w2, d2); d3 = fmaf(x, w3, d3);
        asm volatile("" : "+v"(x));
    }
    out[blockIdx.x * blockDim.x + l] = a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3 + c0 + c1 + c2 + c3 + d0 + d1 + d2 + d3;
}

int main() {
    float* d;
    hipMalloc(&d, 1 << 24);
    k_layout<<<1, 64>>>(d);
    std::vector<float> h(256);
    hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
    // expected under the hypothesis: D[block b][row i][col j] = (1 + i + 10 b) * (100 + j + 1000 b) at lane 4 b + j, register i
    int bad_ij = 0, bad_ji = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int b = l >> 2, j = l & 3;
            const float e1 = (1.f + r + 10.f * b) * (100.f + j + 1000.f * b);      // register = A row, lane%4 = B column
            const float e2 = (1.f + j + 10.f * b) * (100.f + r + 1000.f * b);      // transposed
            bad_ij += h[l * 4 + r] != e1;
            bad_ji += h[l * 4 + r] != e2;
        }
    printf("layout: reg=A-row,lane%%4=B-col mismatches %d ; transposed mismatches %d ; lane 5: %g %g %g %g\n", bad_ij, bad_ji, h[20], h[21], h[22], h[23]);
    const int iters = 4096, blocks = 256 * 4, threads = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 3; ++which) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (which == 0) k_rate<<<blocks, threads>>>(d, iters);
            else if (which == 1) k_rate_dep<<<blocks, threads>>>(d, iters);
            else k_rate_fma<<<blocks, threads>>>(d, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double waves = (double)blocks * threads / 64, n = which == 2 ? 16.0 * iters : 4.0 * iters;
        const double macs = which == 2 ? waves * n * 64 : waves * n * 256;
        printf("%s: %.3f ms, %.1f TMAC/s, %.1f SIMD-cycles per wave-instruction at 2.4 GHz\n", which == 0 ? "mfma 4x4x1 x4 independent" : which == 1 ? "mfma 4x4x1 dependent chain" : "v_fma x16",
               ms, macs / ms / 1e9, ms * 1e-3 * 2.4e9 / (waves * n / 1024.0 / 1.0) / 1.0);
    }
    return 0;
}
