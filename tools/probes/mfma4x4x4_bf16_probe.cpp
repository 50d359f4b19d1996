// Operand layout probe of v_mfma_f32_4x4x4_16b_bf16 (16 independent 4x4x4 blocks per wave).
// Hypothesis: lane l -> block l/4; A: row l%4, its four values are k = 0..3; B: column l%4, four values k = 0..3;
// D: lane = (block, column l%4), register r = row.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const s16x4* a, const s16x4* b, f32x4* d) {
    d[threadIdx.x] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a[threadIdx.x], b[threadIdx.x], f32x4{0, 0, 0, 0}, 0, 0, 0);
}
static unsigned short bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }
int main() {
    float A[16][4][4], B[16][4][4];                       // [block][row][k], [block][k][col]: small integers, exact in bf16
    for (int blk = 0; blk < 16; ++blk)
        for (int i = 0; i < 4; ++i)
            for (int kk = 0; kk < 4; ++kk) { A[blk][i][kk] = (float)((blk * 7 + i * 3 + kk) % 11 - 5); B[blk][kk][i] = (float)((blk * 5 + kk * 2 + i * 3) % 13 - 6); }
    unsigned short ha[64][4], hb[64][4]; float hd[64][4];
    for (int l = 0; l < 64; ++l)
        for (int kk = 0; kk < 4; ++kk) { ha[l][kk] = bf(A[l / 4][l % 4][kk]); hb[l][kk] = bf(B[l / 4][kk][l % 4]); }
    s16x4 *da, *db; f32x4* dd;
    hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dd, sizeof hd);
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
    hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            float want = 0;
            for (int kk = 0; kk < 4; ++kk) want += A[l / 4][r][kk] * B[l / 4][kk][l % 4];
            if (hd[l][r] != want) ++bad;
        }
    printf("4x4x4 bf16: %d mismatches vs hypothesis\n", bad);
    return 0;
}
