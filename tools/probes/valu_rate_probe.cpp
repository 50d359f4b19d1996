// Issue cost of common gfx950 vector-ALU instructions: 4 workgroups x 4 waves per CU (4 waves per SIMD), 16 independent destination
// registers per wave, unrolled; cycles per wave-instruction per SIMD at the 2.4 GHz the PMC passes report.
//   hipcc --offload-arch=gfx950 -O3 -x hip valu_rate_probe.cpp -o valu_rate_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define BODY16(STR, ...)                                                         \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(STR : "+v"(r[i]) : __VA_ARGS__);

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    typename std::conditional<MODE == 37, double, float>::type r[16];
    for (int i = 0; i < 16; ++i) r[i] = threadIdx.x * 0.002f + i;
    unsigned ua = __float_as_uint(a), ub = 3;
    unsigned long mask = 0x5555555555555555ul;
    double pa = __hiloint2double(ua, ua);
    if (MODE == 42 || MODE == 45) asm volatile("s_mov_b64 vcc, %0" :: "s"(mask) : "vcc");
    if (MODE == 41) asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(a), "v"(b) : "vcc");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
            if (MODE == 0) { BODY16("v_fma_f32 %0, %1, %2, %0", "v"(a), "v"(b)) }
            if (MODE == 1) { BODY16("v_add_f32 %0, %1, %0", "v"(a)) }
            if (MODE == 2) { BODY16("v_mul_f32 %0, %1, %0", "v"(a)) }
            if (MODE == 3) { BODY16("v_mov_b32 %0, %1", "v"(a)) }
            if (MODE == 4) { BODY16("v_add_u32 %0, %1, %0", "v"(ua)) }
            if (MODE == 5) { BODY16("v_and_b32 %0, %1, %0", "v"(ua)) }
            if (MODE == 6) { BODY16("v_lshlrev_b32 %0, %1, %0", "v"(ub)) }
            if (MODE == 7) { BODY16("v_exp_f32 %0, %0", "v"(a)) }
            if (MODE == 8) { BODY16("v_cvt_pk_bf16_f32 %0, %0, %1", "v"(a)) }
            if (MODE == 9) { BODY16("v_cndmask_b32 %0, %0, %1, vcc", "v"(a)) }
            if (MODE == 10) { BODY16("v_mul_lo_u32 %0, %0, %1", "v"(ua)) }
            if (MODE == 11) { BODY16("v_max_f32 %0, %1, %0", "v"(a)) }
            if (MODE == 12) { BODY16("v_med3_f32 %0, %0, %1, %2", "v"(a), "v"(b)) }
            if (MODE == 13) { BODY16("v_mad_u32_u24 %0, %0, %1, %2", "v"(ua), "v"(ub)) }
            if (MODE == 14) { BODY16("v_add3_u32 %0, %0, %1, %2", "v"(ua), "v"(ub)) }
            if (MODE == 15) { BODY16("v_perm_b32 %0, %0, %1, %2", "v"(ua), "v"(ub)) }
            if (MODE == 16) { BODY16("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v"(ua)) }
            if (MODE == 17) { BODY16("v_fmac_f32 %0, %1, %2", "v"(a), "v"(b)) }
            if (MODE == 18) { BODY16("v_fma_f32 %0, %1, %2, %0", "s"(a), "v"(b)) }
            if (MODE == 19) { BODY16("v_cvt_f32_bf16 %0, %0", "v"(a)) }
            if (MODE == 20) { BODY16("v_lshl_or_b32 %0, %0, %1, %2", "v"(ub), "v"(ua)) }
            if (MODE == 21) { BODY16("v_bfe_u32 %0, %0, %1, %2", "v"(ub), "v"(ub)) }
            if (MODE == 22) { BODY16("v_cndmask_b32_e64 %0, %0, %1, %2", "v"(a), "s"(mask)) }
            if (MODE == 23) { BODY16("v_sub_f32 %0, %1, %0", "v"(a)) }
            if (MODE == 24) { BODY16("v_min_f32 %0, %1, %0", "v"(a)) }
            if (MODE == 25) { BODY16("v_or_b32 %0, %1, %0", "v"(ua)) }
            if (MODE == 26) { BODY16("v_mul_f32 %0, 0x3fb8aa3b, %0", "v"(a)) }
            if (MODE == 27) { BODY16("v_add_f32 %0, 1.0, %0", "v"(a)) }
            if (MODE == 28) { BODY16("v_mul_f32 %0, %1, %0", "s"(a)) }
            if (MODE == 29) { BODY16("v_lshl_add_u32 %0, %0, 2, %1", "v"(ua)) }
            if (MODE == 30) { BODY16("v_cvt_f32_i32 %0, %0", "v"(ua)) }
            if (MODE == 31) { BODY16("v_accvgpr_write_b32 a0, %0", "v"(ua)) }
            if (MODE == 32) { BODY16("v_readlane_b32 s20, %0, 3", "v"(ua)) }
            if (MODE == 33) { BODY16("v_cmp_gt_f32 vcc, %0, %1", "v"(a)) }
            if (MODE == 34) { BODY16("v_cmp_gt_f32_e64 s[20:21], %0, %1", "v"(a)) }
            if (MODE == 35) { BODY16("v_xor_b32 %0, %1, %0", "v"(ua)) }
            if (MODE == 36) { BODY16("v_sub_u32 %0, %1, %0", "v"(ua)) }
            if (MODE == 37) { BODY16("v_pk_add_f32 %0, %1, %0", "v"(pa)) }
            if (MODE == 38) { BODY16("v_ashrrev_i32 %0, 3, %0", "v"(ua)) }
            if (MODE == 39) { BODY16("v_mul_u32_u24 %0, %1, %0", "v"(ua)) }
            if (MODE == 40) { BODY16("v_fma_f32 %0, %0, %1, 1.0", "v"(a)) }
            if (MODE == 42) { BODY16("v_cndmask_b32 %0, %0, %1, vcc", "v"(a)) }
            if (MODE == 43) { BODY16("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc", "v"(a)) }
            if (MODE == 44) { BODY16("v_cmp_gt_f32_e64 s[20:21], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]", "v"(a)) }
            if (MODE == 45) { BODY16("v_cndmask_b32_e64 %0, %1, %0, vcc", "v"(a)) }
            if (MODE == 41) { BODY16("v_cndmask_b32 %0, %0, %1, vcc", "v"(a)) }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += (float)r[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE> void run(const char* name) {
    float* out; (void)hipMalloc(&out, 256 * 4096 * 4);
    const int iters = 1000, grid = 256 * 4;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<grid, 256>>>(out, 10, 1.0001f, 0.5f);
    (void)hipEventRecord(e0);
    k<MODE><<<grid, 256>>>(out, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double winst = (double)grid * 4 * iters * 8 * 16;
    printf("%-22s %.3f ms  %.2f cycles per wave-instruction per SIMD\n", name, ms, ms * 1e-3 * 2.4e9 / (winst / 1024));
    (void)hipFree(out);
}
int main() {
    run<0>("v_fma_f32"); run<17>("v_fmac_f32"); run<18>("v_fma_f32 sgpr src"); run<1>("v_add_f32"); run<2>("v_mul_f32"); run<11>("v_max_f32");
    run<12>("v_med3_f32"); run<3>("v_mov_b32"); run<4>("v_add_u32"); run<14>("v_add3_u32"); run<5>("v_and_b32"); run<6>("v_lshlrev_b32");
    run<20>("v_lshl_or_b32"); run<21>("v_bfe_u32"); run<13>("v_mad_u32_u24"); run<10>("v_mul_lo_u32"); run<9>("v_cndmask_b32"); run<15>("v_perm_b32");
    run<16>("v_mov_b32 dpp quad"); run<7>("v_exp_f32"); run<8>("v_cvt_pk_bf16_f32"); run<19>("v_cvt_f32_bf16");
    run<22>("v_cndmask_b32_e64 sgpr"); run<41>("v_cndmask_b32 vcc set"); run<23>("v_sub_f32"); run<24>("v_min_f32"); run<25>("v_or_b32"); run<35>("v_xor_b32");
    run<36>("v_sub_u32"); run<26>("v_mul_f32 literal"); run<27>("v_add_f32 inline 1.0"); run<28>("v_mul_f32 sgpr"); run<29>("v_lshl_add_u32");
    run<30>("v_cvt_f32_i32"); run<31>("v_accvgpr_write_b32"); run<32>("v_readlane_b32"); run<33>("v_cmp_gt_f32 vcc"); run<34>("v_cmp_gt_f32_e64 sgpr");
    run<42>("v_cndmask vcc (s_mov)"); run<45>("v_cndmask_e64 vcc"); run<43>("v_cmp + v_cndmask vcc (2 instr)"); run<44>("v_cmp_e64 + cndmask_e64 (2)");
    run<37>("v_pk_add_f32"); run<38>("v_ashrrev_i32"); run<39>("v_mul_u32_u24"); run<40>("v_fma_f32 inline const");
    return 0;
}
