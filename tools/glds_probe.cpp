// Probe: semantics of __builtin_amdgcn_global_load_lds (16 B per lane) on gfx950.
// Each wave copies 1 KiB pieces global -> LDS with a per-lane source (some lanes point at a zero buffer),
// then the workgroup reads LDS back.  Host checks the image.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ float4 g_zero16;

__device__ __forceinline__ void glds16(const float* gsrc, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__global__ void probe(const float* __restrict__ src, float* __restrict__ dst, int n_pieces) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int j = wave; j < n_pieces; j += 4) {
        // piece j: lane l copies source chunk perm(j, l); every 5th lane reads zeros instead
        const int q = j * 64 + ((lane * 7 + j) & 63);
        const float* s = ((lane % 5) == 4) ? reinterpret_cast<const float*>(&g_zero16) : src + 4 * q;
        glds16(s, lds + j * 256);
    }
    __syncthreads();
    for (int i = tid; i < n_pieces * 256; i += 256) dst[i] = lds[i];
}

int main() {
    const int n_pieces = 16, n = n_pieces * 256;
    std::vector<float> h(n), out(n);
    for (int i = 0; i < n; ++i) h[i] = 1.0f + i;
    float *d_src, *d_dst;
    hipMalloc(&d_src, n * 4); hipMalloc(&d_dst, n * 4);
    hipMemcpy(d_src, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(d_dst, 0xff, n * 4);
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), n * 4, 0, d_src, d_dst, n_pieces);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(out.data(), d_dst, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int j = 0; j < n_pieces; ++j)
        for (int l = 0; l < 64; ++l)
            for (int k = 0; k < 4; ++k) {
                const int q = j * 64 + ((l * 7 + j) & 63);
                const float want = (l % 5 == 4) ? 0.f : h[4 * q + k];
                if (out[j * 256 + l * 4 + k] != want) { if (bad < 5) printf("mismatch piece %d lane %d k %d: got %f want %f\n", j, l, k, out[j*256+l*4+k], want); ++bad; }
            }
    printf("glds probe: err=%d bad=%d of %d\n", (int)e, bad, n);
    return bad != 0;
}
