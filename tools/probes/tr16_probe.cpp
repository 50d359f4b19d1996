// Semantics probe of ds_read_b64_tr_b16 (gfx950): which 16-bit elements reach lane l, element j, for per-lane addresses.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/tr16_probe.cpp -o tools/probes/tr16_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out, const int* addr) {
    extern __shared__ __align__(16) unsigned char smem[];
    for (int i = threadIdx.x; i < 8192; i += 64) ((short*)smem)[i] = (short)i;
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + addr[threadIdx.x]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
    int h_addr[64]; short h_out[256];
    int* d_addr; short* d_out;
    hipMalloc(&d_addr, sizeof h_addr); hipMalloc(&d_out, sizeof h_out);
    for (int variant = 0; variant < 2; ++variant) {
        // variant 0: lane l -> byte 8 l (the documented standard image); variant 1: rows scattered: lane l -> row (l>>2) at
        // byte 200 * (l >> 2), 8-byte piece l & 3
        for (int l = 0; l < 64; ++l) h_addr[l] = variant == 0 ? 8 * l : 200 * (l >> 2) + 8 * (l & 3);
        hipMemcpy(d_addr, h_addr, sizeof h_addr, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 16384, 0, d_out, d_addr);
        hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
                const int src_lane = (l & 48) + 4 * j + ((l & 15) >> 2);          // hypothesis: row j of the group's block
                const int want = h_addr[src_lane] / 2 + (l & 3);
                if (h_out[l * 4 + j] != (short)want) ++bad;
            }
        printf("variant %d: %d mismatches vs hypothesis out[l][j] = elem (l&3) of lane (l&48) + 4j + ((l&15)>>2)\n", variant, bad);
        if (bad) for (int l = 0; l < 20; ++l) printf("  lane %2d: %d %d %d %d\n", l, h_out[l*4], h_out[l*4+1], h_out[l*4+2], h_out[l*4+3]);
    }
    return 0;
}
