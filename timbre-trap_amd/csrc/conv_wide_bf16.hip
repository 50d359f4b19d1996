// bf16-STORAGE residual blocks of the wide levels (C = 16, 32; reference modules.py:721-777) for gfx950.
//
// The fp32 kernels of conv_mfma.hip keep every activation as fp32, channel-planar (B,C,H,T); their bf16 modes only round
// the matrix operands, so nothing in HBM gets narrower.  Here the activations of a level live in HBM as bf16,
// CHANNEL-INNERMOST  [B][H][T][C]:
//   * one pixel's channels are one contiguous 32 / 64 bytes, so the K = channels slice of an operand of
//     v_mfma_f32_16x16x32_bf16 (8 bf16 per lane) is ONE 16-byte LDS read (ds_read_b128) -- no conversion, no packing;
//   * a tile row with its halo is one contiguous run in HBM and goes to LDS by LDS-DMA (global_load_lds_dwordx4),
//     out-of-image pieces sourced from a zero page ('same' padding);
//   * the accumulator layout of the 3x3 product (lane = pixel, registers = channels) IS the B-operand layout of the 1x1
//     product, so ELU(conv3x3) feeds conv1x1 from registers, and every epilogue access is 16 bytes per lane.
// All weights are fp32 in HBM (master copy) and are rounded to bf16 into REGISTERS once per workgroup; products accumulate
// in fp32; bias, ELU and the residual add are fp32; stored activations and activation gradients are bf16 (round to nearest
// even).  Weight and bias gradients are fp32.
//
// Kernels (all persistent, XCD-ordered tile walk):
//   k_wide_pack / k_wide_unpack    fp32 planar <-> bf16 channel-innermost at the two ends of a level
//   k_wrb_conv<C,D,0>              y = ELU(W2 . ELU(W1 (*)_D x + b1) + b2) + x, optionally saving h1 = ELU(W1 (*) x + b1)
//   k_wrb_conv<C,D,1>              dx = dy + W1^T (*)_D dA1   (same main loop, weights transposed and taps reversed)
//   k_wrb_bwd_a<C>                 pointwise chain of the backward: a2 = W2 h1 + b2, dA2 = dy ELU'(a2), dh1 = W2^T dA2,
//                                  dA1 = dh1 ELU'(a1); dW2, db1, db2 partials
//   k_wrb_wgrad<C,D>               dW1[co][ci][tap] = sum_pix dA1[co][pix] x[ci][pix + tap]: K = pixels, both operands by
//                                  LDS transpose reads (ds_read_b64_tr_b16) from channel-innermost images
//   k_wrb_reduce<C>                sums the per-wave register dumps into the fp32 gradients (+=) in a fixed order: no atomics at the
//                                  wide levels (the narrow levels' k_nrb_reduce below meets the 16 / C diagonal contributions of a
//                                  dW1 element in atomicAdd: run-to-run differences at fp32 rounding, tests/test_gpu_determinism.py)
#include "wide_common.h"

namespace {

// ---- layout change at the ends of a level --------------------------------------------------------------------------
// One thread per 16-byte piece: eight channels of one pixel (C >= 8) or two whole pixels (C = 4; T even keeps a pair inside
// one image row).
template <int C>
__global__ __launch_bounds__(NT) void k_wide_pack(const float* __restrict__ x, e16* __restrict__ out, int H, int T, long npix) {
    constexpr int CPP = C >= 8 ? 8 : C, PPP = 8 / CPP, CG = C / CPP;         // channels / pixels per piece, pieces per pixel
    const long i = (long)blockIdx.x * NT + threadIdx.x;
    const long pix = (i / CG) * PPP;
    if (pix >= npix) return;
    const int cg = (int)(i % CG);
    const long plane = (long)H * T;
    const long b = pix / plane, o = pix - b * plane;
    const float* s = x + (b * C + cg * CPP) * plane + o;
    e16x8 q;
#pragma unroll
    for (int p = 0; p < PPP; ++p)
#pragma unroll
        for (int j = 0; j < CPP; ++j) q[p * CPP + j] = (e16)s[j * plane + p];
    *reinterpret_cast<e16x8*>(out + pix * C + cg * CPP) = q;
}
template <int C>
__global__ __launch_bounds__(NT) void k_wide_unpack(const e16* __restrict__ in, float* __restrict__ y, int H, int T, long npix) {
    constexpr int CPP = C >= 8 ? 8 : C, PPP = 8 / CPP, CG = C / CPP;
    const long i = (long)blockIdx.x * NT + threadIdx.x;
    const long pix = (i / CG) * PPP;
    if (pix >= npix) return;
    const int cg = (int)(i % CG);
    const long plane = (long)H * T;
    const long b = pix / plane, o = pix - b * plane;
    const e16x8 q = *reinterpret_cast<const e16x8*>(in + pix * C + cg * CPP);
    float* d = y + (b * C + cg * CPP) * plane + o;
#pragma unroll
    for (int p = 0; p < PPP; ++p)
#pragma unroll
        for (int j = 0; j < CPP; ++j) d[j * plane + p] = (float)q[p * CPP + j];
}

// ---- 3x3 main loop: block forward and data gradient ------------------------------------------------------------------
// Sixteen lanes of one k-group read the same 16-byte channel group of sixteen consecutive pixels (64 / 32 bytes apart), which
// would put them on 4 / 8 different bank quads only.  The LDS image therefore stores channel group cg of column `col` at
// position cg ^ cswz(col) of the pixel (a permutation of the DMA sources): any sixteen consecutive columns then cover all
// sixteen bank quads (PMC before: SQ_LDS_BANK_CONFLICT = 4 x SQ_ACTIVE_INST_LDS).
#ifndef TT_H1_NT
#define TT_H1_NT 1
#endif
// The hidden activation is written for the BACKWARD pass, tens of milliseconds away: a non-temporal store keeps it from displacing the block's
// output -- which the next kernel reads at once -- in the L2 / memory-side cache (TT_H1_NT, round 6: 50.97 / 50.74 / 50.68 -> 50.45 / 50.39 / 50.37 ms
// per step, same box, profiles/r06_cache_hints_ab.txt; the isolated call does not change -- it is a cache effect between kernels)
template <class V> __device__ __forceinline__ void h1_store(V* p, const V& v) {
    if (TT_H1_NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
#ifndef TT_CSWZ_NEW
#define TT_CSWZ_NEW 1
#endif
// (round 6: ds_read_b128's lane groups are {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32 -- not runs of sixteen lanes: with the round 2-5 placement
//  ((col >> 2) & 3 / (col >> 3) & 1) a B-operand read took 7-8 LDS cycles instead of 4 (PMC: SQ_LDS_BANK_CONFLICT = 47-50 % of SQ_LDS_IDX_ACTIVE in
//  k_wrb_conv).  TT_CSWZ_NEW (default) is the conflict-free placement for those groups, piece ^= 2 * bit 2 of the column (C = 32) / bit 2 (C = 16):
//  conflicts 9.6 M -> 0 (C = 32), 10.9 M -> 1.1 M (C = 16) per launch, LDS-active cycles halved -- and the SAME time, isolated and in the step
//  (profiles/r06_bwds_ablation.txt): the forward kernels were never bound by their LDS reads.  Shipped because it is right, not because it is faster.)
template <int C> __device__ __forceinline__ int cswz(int col) {
    if (TT_CSWZ_NEW) return C == 32 ? (((col >> 2) & 1) << 1) : ((col >> 2) & 1);
    return C == 32 ? ((col >> 2) & 3) : ((col >> 3) & 1);
}
// Measured at the bench shape (B 64, C 32, H 65, T 1024; 0.20 ms per launch = 4.1 TB/s over x + y + h1):
//   staging alone 0.066 ms, products + epilogue + stores alone 0.134 ms, and they add up -- but an explicit double-buffered tile
//   ring (8 waves, next tile's DMA issued before this tile's products, exact-count vmcnt waits so the stores keep draining)
//   changed nothing (0.20 ms), and neither did the bank swizzle above; without the two ELUs the launch takes 0.178 ms.
//   PMC: FETCH x2 + WRITE = 0.85-0.91 GB against 0.82 GB of x + y + h1 -- no wasted traffic left to remove.

template <int C, int D> struct WT {
    static constexpr int TH = 8, TW = 64;
    static constexpr int CG = C / 8;
    static constexpr int RW = TW + 2 * D, ROWS = TH + 2 * D;
    static constexpr int NP = ROWS * RW * CG;                   // 16-byte pieces of the tile with its halo
    static constexpr int NPR = (NP + NT - 1) / NT * NT;         // whole DMA instructions for every wave
    static constexpr int LDS_BYTES = NPR * 16;
};

// MODE 0: x -> y (and h1 if SAVE).  MODE 1: x = dA1, res = dy, y = dx; b1 / w2 / b2 unused.
// MODE 2 (round 6): MODE 0 with a weighted skip join in the epilogue -- y = block(x) + jw[0] * res[b mod jB], res = an encoder embedding of
// jB clips (reference modules.py:112, :569-589: the join behind a DecoderBlock; b mod jB: both halves of a pair decode take the same
// embedding).  One more 16- / 8-byte load per lane, requested before the products like MODE 1's dy; one rounding of the joined value.
// MINW = waves per SIMD the register allocation must allow: at C = 32 the forward holds 72 + 8 VGPRs of weights and lands
// at 172-176 registers -- two workgroups per CU -- unless capped at 168 (MINW = 3: four registers spill to scratch).
template <int C, int D, int MODE, bool SAVE, int MINW>
__global__ __launch_bounds__(NT, MINW) void k_wrb_conv(const e16* __restrict__ x, const float* __restrict__ w1,
                                                 const float* __restrict__ b1, const float* __restrict__ w2,
                                                 const float* __restrict__ b2, const e16* __restrict__ res,
                                                 e16* __restrict__ y, e16* __restrict__ h1, int B, int H, int T,
                                                 int tiles_h, int tiles_t, int ntiles, const float* __restrict__ jw, int jB) {
    using G = WT<C, D>;
    constexpr bool FWD = MODE != 1, JOIN = MODE == 2;
    float jsc = 0.f;
    if constexpr (JOIN) jsc = jw ? jw[0] : 1.f;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    constexpr int NCT = C / 16;                                  // co-tiles
    constexpr int NK = C == 32 ? 9 : 5;                          // products per co-tile: one tap (C = 32) / two taps (C = 16)
    constexpr int NCH = C == 32 ? 8 : 4;                         // channels a lane ends up with

    // ---- weights to registers, rounded to bf16 ----
    e16x8 A[NK][NCT];
    float chk = 0.f;                                             // forward: non-finite parameters (poison_acc, bf16_common.h)
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int tap, kc;                                     // tap and contraction channel of this lane's k = 8 g + j
                if (C == 32) { tap = k; kc = 8 * g + j; }
                else { tap = 2 * k + (g >> 1); kc = 8 * (g & 1) + j; }
                const int mo = chan_of<C>(ct, n);                // output channel of row n
                float wv = 0.f;
                if (tap < 9) wv = FWD ? w1[(mo * C + kc) * 9 + tap] : w1[(kc * C + mo) * 9 + (8 - tap)];
                v[j] = wv;
                if (FWD) chk = poison_acc(chk, wv);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) A[k][ct][j] = (e16)v[j];
        }
    e16x8 A2[NCT];                                              // C = 32: W2 rows (K = 32)
    s16x4 A2s;                                                   // C = 16: W2 rows (K = 16)
    float b1r[NCH], b2r[NCH];
    if constexpr (FWD) {
        if constexpr (C == 32) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float wv = w2[chan_of<C>(ct, n) * C + 8 * g + j];
                    chk = poison_acc(chk, wv);
                    A2[ct][j] = (e16)wv;
                }
        } else {
            e16x4 t;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float wv = w2[n * C + 4 * g + j]; chk = poison_acc(chk, wv); t[j] = (e16)wv; }
            A2s = __builtin_bit_cast(s16x4, t);
        }
#pragma unroll
        for (int j = 0; j < NCH; ++j) { b1r[j] = b1[NCH * g + j]; b2r[j] = b2[NCH * g + j]; chk = poison_acc(poison_acc(chk, b1r[j]), b2r[j]); }
    }
    // a NaN / inf weight or bias must reach the output like in torch (the ELU forms below return 0 for a NaN pre-activation): the
    // workgroup then writes NaN to every output it owns
    const bool poisoned = FWD && params_poisoned(chk);          // (a non-finite join weight propagates through the fma by itself)

    const e16* zero = reinterpret_cast<const e16*>(&g_wzero16);
    for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
        int tile = xcd_order(v, ntiles);
        // JOIN over two batches that share the embedding (B = 2 jB): the two halves' tiles of the same (clip, rows, columns) are NEIGHBOURS in
        // the walk, so that the second read of the embedding's pixels finds them in the XCD's L2 (the half is the fastest index)
        int half = 0;
        if (JOIN && B == 2 * jB) { half = tile & 1; tile >>= 1; }
        const int tt = tile % tiles_t; tile /= tiles_t;
        const int th = tile % tiles_h;
        const int b = tile / tiles_h + half * jB, h0 = th * G::TH, t0 = tt * G::TW;
        const e16* xb = x + (long)b * H * T * C;

        __syncthreads();                                         // the previous tile has been consumed
        for (int i = wave * 64; i < G::NPR; i += NT) {
            const int p = i + lane;
            const int row = p / (G::RW * G::CG), rem = p - row * (G::RW * G::CG);
            const int px = rem / G::CG, cg = (rem - px * G::CG) ^ cswz<C>(px);      // 16-byte position -> channel group held there
            const int h = h0 - D + row, t = t0 - D + px;
            const bool ok = p < G::NP && (unsigned)h < (unsigned)H && (unsigned)t < (unsigned)T;
            glds16(ok ? xb + ((long)h * T + t) * C + cg * 8 : zero, smem + (long)i * 16);
        }
        const int c0 = wave * 16;
        const int t = t0 + c0 + n;
        // JOIN: the embedding's pixels one row ahead of their use -- the first row's request rides on the staging wait below, row r + 1's is
        // issued at the top of row r (requested inside the row that uses it, every row exposed a full memory latency: +130-160 us per launch)
        typename std::conditional<C == 32, e16x8, e16x4>::type jq;
        const e16* jrow = nullptr;
        if constexpr (JOIN) {
            jrow = res + (((long)(b % jB) * H) * T + (t < T ? t : T - 1)) * C + NCH * g;
            jq = *reinterpret_cast<const decltype(jq)*>(jrow + (long)(h0 < H ? h0 : H - 1) * T * C);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        // (Round 6, measured and dropped: the row loop unrolled by two at C = 16 -- two independent chains of five matrix products per body: isolated
        //  0.208 / 0.208 / 0.206 -> 0.195 / 0.205 / 0.203 ms, train step 51.33 / 51.01 / 50.97 -> 51.16 / 51.03 / 51.00 ms: nothing.)
        for (int r = 0; r < G::TH; ++r) {
            const int h = h0 + r;
            if (h >= H) break;
            const long pix = ((long)b * H + h) * T + t;
            const bool valid = t < T;
            if (poisoned) {
                if (valid) {
                    typename std::conditional<C == 32, e16x8, e16x4>::type qn;
#pragma unroll
                    for (int j = 0; j < NCH; ++j) qn[j] = (e16)__builtin_nanf("");
                    *reinterpret_cast<decltype(qn)*>(y + pix * C + NCH * g) = qn;
                    if (SAVE) *reinterpret_cast<decltype(qn)*>(h1 + pix * C + NCH * g) = qn;
                }
                continue;
            }
            f32x4 acc[NCT];                                      // the bias enters as the accumulator's initial value
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                if constexpr (FWD) acc[ct] = f32x4{b1r[4 * ct], b1r[4 * ct + 1], b1r[4 * ct + 2], b1r[4 * ct + 3]};
                else acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            typename std::conditional<C == 32, e16x8, e16x4>::type rq;      // MODE 1: dy of this pixel, requested early
            if constexpr (MODE == 1)            // unconditional (clamped) so that no branch pins a wait in front of the products
                rq = *reinterpret_cast<const decltype(rq)*>(res + (valid ? pix : pix - (t - (T - 1))) * C + NCH * g);
            if constexpr (JOIN) {               // the embedding's pixel of clip b mod jB: this row's arrived a row ago, the next row's goes out
                rq = jq;
                const int hn = h + 1 < H ? h + 1 : H - 1;
                jq = *reinterpret_cast<const decltype(jq)*>(jrow + (long)hn * T * C);
            }
            e16x8 centre;
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                int tap = C == 32 ? k : 2 * k + (g >> 1);
                if (tap > 8) tap = 8;                            // the weights of the missing tenth tap are zero
                const int kh = tap / 3, kw = tap - 3 * kh;
                const int col = c0 + n + kw * D, pxi = (r + kh * D) * G::RW + col;
                const int cho = 8 * ((C == 32 ? g : (g & 1)) ^ cswz<C>(col));
                const e16x8 bq = *reinterpret_cast<const e16x8*>(smem + ((long)pxi * C + cho) * 2);
                if (C == 32 && k == 4) centre = bq;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) acc[ct] = mma32(A[k][ct], bq, acc[ct]);
            }
            float val[NCH];
            if constexpr (C == 32) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { val[j] = acc[0][j]; val[4 + j] = acc[1][j]; }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) val[j] = acc[0][j];
            }
            if constexpr (MODE == 1) {
                if (valid) {
                    if constexpr (C == 32) {
                        e16x8 o;
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = (e16)(val[j] + (float)rq[j]);
                        *reinterpret_cast<e16x8*>(y + pix * C + 8 * g) = o;
                    } else {
                        e16x4 o;
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = (e16)(val[j] + (float)rq[j]);
                        *reinterpret_cast<e16x4*>(y + pix * C + 4 * g) = o;
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < NCH; ++j) val[j] = elu_res(val[j]);
                if constexpr (C == 32) {
                    e16x8 hq;
#pragma unroll
                    for (int j = 0; j < 8; ++j) hq[j] = (e16)val[j];
                    if (SAVE && valid) h1_store(reinterpret_cast<e16x8*>(h1 + pix * C + 8 * g), hq);
                    f32x4 z0 = mma32(A2[0], hq, f32x4{b2r[0], b2r[1], b2r[2], b2r[3]});
                    f32x4 z1 = mma32(A2[1], hq, f32x4{b2r[4], b2r[5], b2r[6], b2r[7]});
                    e16x8 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr (JOIN) {
                            o[j] = (e16)__builtin_fmaf(jsc, (float)rq[j], elu_out(z0[j]) + (float)centre[j]);
                            o[4 + j] = (e16)__builtin_fmaf(jsc, (float)rq[4 + j], elu_out(z1[j]) + (float)centre[4 + j]);
                        } else {
                            o[j] = (e16)(elu_out(z0[j]) + (float)centre[j]);
                            o[4 + j] = (e16)(elu_out(z1[j]) + (float)centre[4 + j]);
                        }
                    }
                    if (valid) *reinterpret_cast<e16x8*>(y + pix * C + 8 * g) = o;
                } else {
                    e16x4 hq;
#pragma unroll
                    for (int j = 0; j < 4; ++j) hq[j] = (e16)val[j];
                    if (SAVE && valid) h1_store(reinterpret_cast<e16x4*>(h1 + pix * C + 4 * g), hq);
                    const f32x4 z = mma16(A2s, __builtin_bit_cast(s16x4, hq), f32x4{b2r[0], b2r[1], b2r[2], b2r[3]});
                    const int colc = c0 + n + D, pxc = (r + D) * G::RW + colc;
                    const e16x4 xc = *reinterpret_cast<const e16x4*>(smem + ((long)pxc * C + 8 * ((g >> 1) ^ cswz<C>(colc)) + 4 * (g & 1)) * 2);
                    e16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = JOIN ? (e16)__builtin_fmaf(jsc, (float)rq[j], elu_out(z[j]) + (float)xc[j]) : (e16)(elu_out(z[j]) + (float)xc[j]);
                    if (valid) *reinterpret_cast<e16x4*>(y + pix * C + 4 * g) = o;
                }
            }
        }
    }
}

// ---- pointwise chain of the backward ---------------------------------------------------------------------------------
// One wave per group of 16 consecutive pixels (the tensor is [npix][C]); operands straight from HBM in B-operand layout.
// Every workgroup leaves its accumulators as a raw register dump [dW2: ((a NCT + c) 4 + r) 64 + lane][db1 C][db2 C]
// (the four waves summed through LDS); k_wrb_reduce sums the dumps of all workgroups in a fixed order (no atomics).
// The 16-pixel tiles are read back transposed (lds_tr16, bf16_common.h) as the K = pixels operands of the dW2 product.
template <int C> struct WA {
    static constexpr int PS = C * 2 + 8;                         // bytes per pixel of the transposition buffers (bank spread)
    static constexpr int WAVE_BYTES = 2 * 16 * PS;               // dA2 and h1 of one group
    static constexpr int DUMP = C * C + 2 * C;                   // floats per wave
    static constexpr int LDS_BYTES = 4 * WAVE_BYTES > 16 * DUMP ? 4 * WAVE_BYTES : 16 * DUMP;
};

template <int C>
__global__ __launch_bounds__(NT) void k_wrb_bwd_a(const e16* __restrict__ h1, const e16* __restrict__ dy,
                                                  const float* __restrict__ w2, const float* __restrict__ b2,
                                                  e16* __restrict__ da1, float* __restrict__ part, long npix, long ngroups) {
    using G = WA<C>;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    constexpr int NCT = C / 16;
    constexpr int NCH = C == 32 ? 8 : 4;
    typedef typename std::conditional<C == 32, e16x8, e16x4>::type vec_t;     // a lane's channels of one pixel

    // W2 (rows = output channel) and W2^T (rows = input channel), both with the row order of chan_of
    e16x8 A2[NCT], A2T[NCT];
    s16x4 A2s, A2Ts;
    if constexpr (C == 32) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                A2[ct][j] = (e16)w2[chan_of<C>(ct, n) * C + 8 * g + j];
                A2T[ct][j] = (e16)w2[(8 * g + j) * C + chan_of<C>(ct, n)];
            }
    } else {
        e16x4 a, at;
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] = (e16)w2[n * C + 4 * g + j]; at[j] = (e16)w2[(4 * g + j) * C + n]; }
        A2s = __builtin_bit_cast(s16x4, a); A2Ts = __builtin_bit_cast(s16x4, at);
    }
    float b2r[NCH], db1a[NCH], db2a[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) { b2r[j] = b2[NCH * g + j]; db1a[j] = 0.f; db2a[j] = 0.f; }
    f32x4 dw2[NCT][NCT];
#pragma unroll
    for (int a = 0; a < NCT; ++a)
#pragma unroll
        for (int c = 0; c < NCT; ++c) dw2[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    unsigned char* tg = smem + wave * G::WAVE_BYTES;             // this wave's dA2 tile, then its h1 tile
    unsigned char* thh = tg + 16 * G::PS;
    const int trj = n >> 2, trq = n & 3;                         // transpose read: this lane supplies row trj, columns 4 trq..

    vec_t zero_v;
#pragma unroll
    for (int j = 0; j < NCH; ++j) zero_v[j] = (e16)0.f;
    const long gstride = (long)gridDim.x * 4;
    long grp = (long)blockIdx.x * 4 + wave;
    vec_t hq = zero_v, dq = zero_v;
    if (grp < ngroups && grp * 16 + n < npix) {
        hq = *reinterpret_cast<const vec_t*>(h1 + (grp * 16 + n) * C + NCH * g);
        dq = *reinterpret_cast<const vec_t*>(dy + (grp * 16 + n) * C + NCH * g);
    }
    for (; grp < ngroups; grp += gstride) {
        const long pix = grp * 16 + n;
        const bool valid = pix < npix;
        // next group's operands are on their way while this one is processed
        const long pn = (grp + gstride) * 16 + n;
        vec_t hq_n = zero_v, dq_n = zero_v;
        if (grp + gstride < ngroups && pn < npix) {
            hq_n = *reinterpret_cast<const vec_t*>(h1 + pn * C + NCH * g);
            dq_n = *reinterpret_cast<const vec_t*>(dy + pn * C + NCH * g);
        }
        float hv[NCH], gv[NCH], a1g[NCH];
        vec_t gq, aq;
        f32x4 z[NCT], u[NCT];
        if constexpr (C == 32) {
            z[0] = mma32(A2[0], hq, f32x4{b2r[0], b2r[1], b2r[2], b2r[3]});
            z[1] = mma32(A2[1], hq, f32x4{b2r[4], b2r[5], b2r[6], b2r[7]});
        } else {
            z[0] = mma16(A2s, __builtin_bit_cast(s16x4, hq), f32x4{b2r[0], b2r[1], b2r[2], b2r[3]});
        }
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const float a2 = z[j >> 2][j & 3];
            gv[j] = (float)dq[j] * elu_dpre(a2);
            gq[j] = (e16)gv[j];
            hv[j] = (float)hq[j];
        }
        if constexpr (C == 32) {
            u[0] = mma32(A2T[0], gq, f32x4{0.f, 0.f, 0.f, 0.f});
            u[1] = mma32(A2T[1], gq, f32x4{0.f, 0.f, 0.f, 0.f});
        } else {
            u[0] = mma16(A2Ts, __builtin_bit_cast(s16x4, gq), f32x4{0.f, 0.f, 0.f, 0.f});
        }
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            a1g[j] = u[j >> 2][j & 3] * elu_dout(hv[j]);
            aq[j] = (e16)a1g[j];
            db2a[j] += gv[j]; db1a[j] += a1g[j];                  // invalid pixels contribute zeros
        }
        if (valid) *reinterpret_cast<vec_t*>(da1 + pix * C + NCH * g) = aq;
        // transposition buffers: pixel n, this lane's channels, in 8-byte pieces
        if constexpr (C == 32) {
            const uint2* gs = reinterpret_cast<const uint2*>(&gq);
            const uint2* hs = reinterpret_cast<const uint2*>(&hq);
            uint2* gd = reinterpret_cast<uint2*>(tg + n * G::PS + 16 * g);
            uint2* hd = reinterpret_cast<uint2*>(thh + n * G::PS + 16 * g);
            gd[0] = gs[0]; gd[1] = gs[1]; hd[0] = hs[0]; hd[1] = hs[1];
        } else {
            *reinterpret_cast<uint2*>(tg + n * G::PS + 8 * g) = __builtin_bit_cast(uint2, gq);
            *reinterpret_cast<uint2*>(thh + n * G::PS + 8 * g) = __builtin_bit_cast(uint2, hq);
        }
        // dW2[co][ci] += sum over the 16 pixels dA2[co][p] h1[ci][p]: the tiles read back transposed
        // (same wave wrote them: LDS operations of one wave complete in order; the fence only stops the compiler)
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        s16x4 ga[NCT], hb[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            ga[ct] = lds_tr16(tg + (4 * g + trj) * G::PS + (16 * ct + 4 * trq) * 2);
            hb[ct] = lds_tr16(thh + (4 * g + trj) * G::PS + (16 * ct + 4 * trq) * 2);
        }
#pragma unroll
        for (int a = 0; a < NCT; ++a)
#pragma unroll
            for (int c = 0; c < NCT; ++c) dw2[a][c] = mma16(ga[a], hb[c], dw2[a][c]);
        asm volatile("" ::: "memory");
        hq = hq_n; dq = dq_n;
    }

    // ---- register dumps of the four waves, summed in LDS (plain stores, no atomics), one dump per workgroup ----
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem) + wave * G::DUMP;
#pragma unroll
    for (int a = 0; a < NCT; ++a)
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((a * NCT + c) * 4 + r) * 64 + lane] = dw2[a][c][r];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        float s1 = db1a[j], s2 = db2a[j];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        if (n == 0) { red[C * C + NCH * g + j] = s1; red[C * C + C + NCH * g + j] = s2; }
    }
    __syncthreads();
    const float* all = reinterpret_cast<const float*>(smem);
    float* pw = part + (long)blockIdx.x * G::DUMP;
    for (int i = tid; i < G::DUMP; i += NT) pw[i] = (all[i] + all[G::DUMP + i]) + (all[2 * G::DUMP + i] + all[3 * G::DUMP + i]);
}

// ---- 3x3 weight gradient ---------------------------------------------------------------------------------------------
//   dW1[co][ci][tap] = sum_pix dA1[co][pix] x[ci][pix + tap]     K = pixels, 32 per product (v_mfma_f32_16x16x32_bf16)
// The x tile (with halo) and the dA1 tile are channel-innermost LDS images filled by LDS-DMA; both operands come out of
// them by transpose reads (two per operand: pixels 4g..4g+3 and 16+4g..16+4g+3 of the chunk -- the K order is free as long
// as both operands use the same one).  At C = 32 the two 32-byte halves of a pixel are swapped in every second group of
// four pixels (a permutation of the DMA sources), which keeps the eight pixels one half-wave reads on different banks.
// Wave roles: C = 32: ci-tile w & 1, column half w >> 1 (x is the operand read nine times, so it is the one split);
//             C = 16: column half w & 1, rows of parity w >> 1.
// Every wave dumps its accumulators [(tap NA + a) 4 + r][lane]; k_wrb_reduce sums the dumps.
template <int C, int D> struct WG {
    static constexpr int TH = C == 32 ? 4 : 8, TW = 64;
    static constexpr int CG = C / 8;
    static constexpr int RW = TW + 2 * D, ROWS = TH + 2 * D;
    static constexpr int XP = ROWS * RW * CG, GP = TH * TW * CG;   // 16-byte pieces
    static constexpr int XPR = (XP + NT - 1) / NT * NT;
    static constexpr int X_BYTES = XPR * 16, G_BYTES = GP * 16;
    static constexpr int NA = C / 16;                              // co-tiles a wave accumulates
    static constexpr int DUMP = 9 * NA * 256;                      // floats per wave
    static constexpr int LDS_BYTES = X_BYTES + G_BYTES;
};

template <int C> __device__ __forceinline__ int swz_piece(int q, int cg) { return C == 32 ? (cg ^ (((q >> 2) & 1) << 1)) : cg; }

template <int C, int D>
__global__ __launch_bounds__(NT) void k_wrb_wgrad(const e16* __restrict__ x, const e16* __restrict__ da1,
                                                  float* __restrict__ part, int B, int H, int T, int tiles_h, int tiles_t,
                                                  int ntiles) {
    using G = WG<C, D>;
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* xs = smem;
    unsigned char* gs = smem + G::X_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int trj = n >> 2, trq = n & 3;
    constexpr int NA = G::NA;
    const int cit = C == 32 ? (wave & 1) : 0;                    // ci-tile of this wave
    const int colh = C == 32 ? (wave >> 1) : (wave & 1);         // 32-pixel column half
    const int row0 = C == 32 ? 0 : (wave >> 1), rstep = C == 32 ? 1 : 2;
    f32x4 acc[9][NA];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int a = 0; a < NA; ++a) acc[k][a] = f32x4{0.f, 0.f, 0.f, 0.f};

    const e16* zero = reinterpret_cast<const e16*>(&g_wzero16);
    for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
        int tile = xcd_order(v, ntiles);
        const int tt = tile % tiles_t; tile /= tiles_t;
        const int th = tile % tiles_h;
        const int b = tile / tiles_h, h0 = th * G::TH, t0 = tt * G::TW;
        const e16* xb = x + (long)b * H * T * C;
        const e16* gb = da1 + (long)b * H * T * C;
        __syncthreads();
        for (int i = wave * 64; i < G::XPR; i += NT) {
            const int p = i + lane, q = p / G::CG, cg = swz_piece<C>(q, p - q * G::CG);
            const int row = q / G::RW, px = q - row * G::RW;
            const int h = h0 - D + row, t = t0 - D + px;
            const bool ok = p < G::XP && (unsigned)h < (unsigned)H && (unsigned)t < (unsigned)T;
            glds16(ok ? xb + ((long)h * T + t) * C + cg * 8 : zero, xs + (long)i * 16);
        }
        for (int i = wave * 64; i < G::GP; i += NT) {
            const int p = i + lane, q = p / G::CG, cg = swz_piece<C>(q, p - q * G::CG);
            const int row = q / G::TW, px = q - row * G::TW;
            const int h = h0 + row, t = t0 + px;
            const bool ok = h < H && t < T;
            glds16(ok ? gb + ((long)h * T + t) * C + cg * 8 : zero, gs + (long)i * 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        const int c0 = colh * 32 + 4 * g + trj;                  // first-half pixel whose address this lane supplies
        for (int r = row0; r < G::TH; r += rstep) {
            if (h0 + r >= H) break;
            // dA1 operand(s): co-tiles 0..NA-1
            e16x8 ga[NA];
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                s16x4 lo, hi;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int q = r * G::TW + c0 + 16 * u;
                    const int half = C == 32 ? ((a ^ ((q >> 2) & 1)) << 5) : 0;
                    const s16x4 t4 = lds_tr16(gs + (long)q * (C * 2) + half + trq * 8);
                    if (u == 0) lo = t4; else hi = t4;
                }
                ga[a] = __builtin_bit_cast(e16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int kh = k / 3, kw = k - 3 * kh;
                s16x4 lo, hi;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int q = (r + kh * D) * G::RW + c0 + 16 * u + kw * D;
                    const int half = C == 32 ? ((cit ^ ((q >> 2) & 1)) << 5) : 0;
                    const s16x4 t4 = lds_tr16(xs + (long)q * (C * 2) + half + trq * 8);
                    if (u == 0) lo = t4; else hi = t4;
                }
                const e16x8 xq = __builtin_bit_cast(e16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int a = 0; a < NA; ++a) acc[k][a] = mma32(ga[a], xq, acc[k][a]);
            }
        }
    }

    float* pw = part + ((long)blockIdx.x * 4 + wave) * G::DUMP;
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) pw[((k * NA + a) * 4 + r) * 64 + lane] = acc[k][a][r];
}

// ---- data gradient and weight gradient in ONE pass -----------------------------------------------------------------------
// k_wrb_conv<..,1> and k_wrb_wgrad both stage the dA1 tile; here it is staged once, next to the x tile, in the layout that serves
// both access patterns (fswz, wide_common.h): dx = dy + W1^T (*)_D dA1 over the tile by B-operand reads, then
// dW1[co][ci][tap] += dA1 (x) x(+tap) by transpose reads of the same two images.  dA1 is read from HBM once instead of twice:
// four tensor passes (dA1, dy, x | dx) for what took five, and one launch less.  No halo beyond D, no recomputation -- the vector
// work is the sum of the two kernels minus one staging loop (PMC, round 3: both were parked on memory 40-67 % of their life).
// Round 5: the x tile WITHOUT halo (the halo.d form of rounds 3 / 4 is gone).  dW1[tap] = sum_q x[q] (x) dA1[q - off(tap)] indexes the weight gradient by the pixel
// of x (the identity the strip kernel and k_nrb_bwd_fused use), and dA1 has its D halo anyway: the x image shrinks from
// (TH + 2 D) x (TW + 2 D) to TH x TW pixels -- 25 / 41 / 52 % fewer staged bytes and DMA instructions at dilation 1 / 2 / 3, and at
// C = 32 a third workgroup per CU at dilation 3 (68 -> 50 KB) and 8-row tiles instead of 6 at dilation 2 (55 -> 44 KB).
template <int C, int D, int TH, int TW> struct DXW {
    static constexpr int CG = C / 8, PB = C * 2;
    static constexpr int IR = TH + 2 * D, IW = TW + 2 * D;       // dA1 image: tile + D halo
    static constexpr int NP = IR * IW * CG;
    static constexpr int NPR = (NP + NT - 1) / NT * NT;
    static constexpr int IMG_BYTES = NPR * 16;
    static constexpr int XR = TH, XW = TW;                       // x image: the tile's own pixels
    static constexpr int XNP = XR * XW * CG;
    static constexpr int XNPR = (XNP + NT - 1) / NT * NT;
    static constexpr int X_BYTES = XNPR * 16;
    static constexpr int WDUMP = 9 * (C / 16) * 256;
    // C = 16: the four waves' accumulators are summed through LDS at the end (36.9 KB, more than the two images at dilation 1, 2)
    static constexpr int LDS_BYTES = (C == 16 && IMG_BYTES + X_BYTES < 4 * WDUMP * 4) ? 4 * WDUMP * 4 : IMG_BYTES + X_BYTES;
    static_assert(TW % 32 == 0, "the K = 32 pixels of a weight-gradient product are 32 consecutive columns");
};

// SJ (with GOUT): the backward of a skip join on this block's input rides on the epilogue (SkipJ, wide_common.h)
template <int C, int D, int TH, int TW, bool GOUT = false, bool SJ = false>
__global__ __launch_bounds__(NT, (GOUT && C == 32) ? 3 : 2) void k_wrb_dxw(const e16* __restrict__ x, const e16* __restrict__ da1, const e16* __restrict__ dy,
                                                   const float* __restrict__ w1, e16* __restrict__ dx, float* __restrict__ part_w, int B,
                                                   int H, int T, int tiles_h, int tiles_t, int ntiles, SkipJ sj = SkipJ()) {
    static_assert(!SJ || GOUT, "a skip join rides on the gated epilogue only");
    float sjw = 0.f, sjdot = 0.f;
    if constexpr (SJ) sjw = sj.w ? sj.w[0] : 1.f;
    using G = DXW<C, D, TH, TW>;
    using K = WK<C>;
    constexpr int NCT = K::NCT, NK = K::NK, NCH = K::NCH, PB = G::PB;
    typedef typename std::conditional<C == 32, e16x8, e16x4>::type vec_t;
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* gs = smem;                                    // dA1 (with halo)
    unsigned char* xs = smem + G::IMG_BYTES;                     // x   (the tile's own pixels)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4, trj = n >> 2, trq = n & 3;

    // data-gradient weights (transposed, taps reversed), bf16, in registers for the whole launch
    e16x8 A[NK][NCT];
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int tap, kc;
                if (C == 32) { tap = k; kc = 8 * g + j; }
                else { tap = 2 * k + (g >> 1); kc = 8 * (g & 1) + j; }
                v[j] = tap < 9 ? w1[(kc * C + chan_of<C>(ct, n)) * 9 + (8 - tap)] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) A[k][ct][j] = (e16)v[j];
        }
    // weight-gradient roles (as in k_wrb_bwd_fused): C = 32: wave = (ci-tile wave & 1, co-tile wave >> 1), all rows and chunks;
    // C = 16: the waves split the 32-column chunks, then the rows
    constexpr int NCHK = TW / 32;
    static_assert(C == 32 || (4 % NCHK == 0), "wave roles");
    const int cit = C == 32 ? (wave & 1) : 0, aw = C == 32 ? (wave >> 1) : 0;
    const int ch0 = C == 32 ? 0 : wave % NCHK, rpar = C == 32 ? 0 : wave / NCHK;
    constexpr int CHSTEP = C == 32 ? 1 : NCHK, RSTEP = C == 32 ? 1 : 4 / NCHK;
    f32x4 wacc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wacc[k] = f32x4{0.f, 0.f, 0.f, 0.f};

    const e16* zero = reinterpret_cast<const e16*>(&g_wzero16);
    for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
        int tile = xcd_order(v, ntiles);
        const int tt = tile % tiles_t; tile /= tiles_t;
        const int th = tile % tiles_h;
        const int b = tile / tiles_h, h0 = th * TH, t0 = tt * TW;
        const long ib = (long)b * H * T * C;
        __syncthreads();                                         // the previous tile has been consumed
        for (int i = wave * 64; i < G::NPR; i += NT) {
            const int p = i + lane, q = p / G::CG, s = p - q * G::CG;
            const int row = q / G::IW, px = q - row * G::IW;
            const int h = h0 - D + row, t = t0 - D + px;
            const bool ok = p < G::NP && (unsigned)h < (unsigned)H && (unsigned)t < (unsigned)T;
            const long off = ib + ((long)h * T + t) * C + (s ^ fswz<C>(px)) * 8;
            glds16(ok ? da1 + off : zero, gs + (long)i * 16);
        }
        for (int i = wave * 64; i < G::XNPR; i += NT) {
            const int p = i + lane, q = p / G::CG, s = p - q * G::CG;
            const int row = q / TW, px = q - row * TW;
            const int h = h0 + row, t = t0 + px;
            const bool ok = p < G::XNP && h < H && t < T;
            glds16_x(ok ? x + ib + ((long)h * T + t) * C + (s ^ fswz<C>(px)) * 8 : zero, xs + (long)i * 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        // ---- dx = dy + W1^T (*) dA1 over the tile's own pixels ----
        constexpr int GPRW = TW / 16;
        for (int grp = wave; grp < TH * GPRW; grp += 4) {
            const int r = grp / GPRW, c = (grp - r * GPRW) * 16 + n;
            const int h = h0 + r;
            if (h >= H) break;
            const int t = t0 + c;
            const bool valid = t < T;
            const long pix = ((long)b * H + h) * T + t;
            vec_t sg0, sg1;
            if constexpr (SJ) {                                  // the join's gradient at this pixel, both batches: requested with dy
                const e16* gp = sj.g + (valid ? pix : pix - (t - (T - 1))) * C + NCH * g;
                sg0 = *reinterpret_cast<const vec_t*>(gp);
                sg1 = *reinterpret_cast<const vec_t*>(gp + sj.half);
            }
            // unconditional (clamped) so that no branch pins a wait in front of the products.  (Round 6, measured and dropped: dy requested one
            // pixel group AHEAD of its products, here and in the C = 16 strips' data-gradient phase -- 49.72-49.86 / 49.80-49.86 / 49.85-49.96 ms
            // per step without / strips only / both, profiles/r06_pf_ab.txt: this latency is already hidden by the CU's other waves.)
            const vec_t rq = *reinterpret_cast<const vec_t*>(dy + (valid ? pix : pix - (t - (T - 1))) * C + NCH * g);
            f32x4 acc[NCT];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
            e16x8 unused;
            conv_taps<C, D, G::IW>(gs, r, c, g, A, acc, unused);
            vec_t o;
            if constexpr (GOUT) {
                // dx * ELU'(x) for the layer in front of the level (see k_nrb_bwd_fused): the lane's channels of x from the halo-free image
                // (the x read stays BEHIND the products and the residual: its registers on top of the live dy piece were 176 instead of 168,
                // i.e. two workgroups per CU instead of three -- 0.343 instead of 0.236 ms per call)
                float sum[NCH];
#pragma unroll
                for (int j = 0; j < NCH; ++j) sum[j] = acc[j >> 2][j & 3] + (float)rq[j];
                asm volatile("" ::: "memory");
                const vec_t xq = *reinterpret_cast<const vec_t*>(xs + (r * TW + c) * PB + 16 * ((C == 32 ? g : (g >> 1)) ^ fswz<C>(c)) +
                                                                 (C == 32 ? 0 : 8 * (g & 1)));
                if constexpr (SJ) {
                    float add[NCH], dot = 0.f;
                    skipj_terms<vec_t, NCH>(sg0, sg1, sj.half != 0, xq, sjw, add, dot);
                    if (valid) sjdot += dot;
#pragma unroll
                    for (int j = 0; j < NCH; ++j) sum[j] += add[j];
                }
#pragma unroll
                for (int j = 0; j < NCH; ++j) o[j] = (e16)(sum[j] * elu_dout((float)xq[j]));
            } else {
#pragma unroll
                for (int j = 0; j < NCH; ++j) o[j] = (e16)(acc[j >> 2][j & 3] + (float)rq[j]);
            }
            if (valid) *reinterpret_cast<vec_t*>(dx + pix * C + NCH * g) = o;
        }

        // ---- dW1, K = pixels: 32 consecutive columns of a row per product, both operands by transpose reads ----
        for (int r = rpar; r < TH; r += RSTEP) {
            if (h0 + r >= H) break;
#pragma unroll
            for (int ch = ch0; ch < NCHK; ch += CHSTEP) {
                s16x4 lo, hi;
                // indexed by the pixel of x: B = x at (r, 32 ch ..) of its halo-free image, A = dA1 at row r + (2 - kh) D, column (2 - kw) D + ..
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int xc = ch * 32 + 4 * g + trj + 16 * u;
                    const s16x4 t4 = lds_tr16(xs + (r * TW + xc) * PB + 16 * (((C == 32 ? 2 * cit : 0) + (trq >> 1)) ^ fswz<C>(xc)) + 8 * (trq & 1));
                    if (u == 0) lo = t4; else hi = t4;
                }
                const e16x8 xq = __builtin_bit_cast(e16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const int kh = k / 3, kw = k - 3 * kh;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int cc = (2 - kw) * D + ch * 32 + 4 * g + trj + 16 * u;
                        const s16x4 t4 = lds_tr16(gs + ((r + (2 - kh) * D) * G::IW + cc) * PB +
                                                  16 * (((C == 32 ? 2 * aw : 0) + (trq >> 1)) ^ fswz<C>(cc)) + 8 * (trq & 1));
                        if (u == 0) lo = t4; else hi = t4;
                    }
                    wacc[k] = mma32(__builtin_bit_cast(e16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7)), xq, wacc[k]);
                }
            }
        }
    }
    if constexpr (SJ) skipj_finish(sjdot, sj);
    if constexpr (C == 32) {
        float* pw = part_w + ((long)blockIdx.x * 4 + wave) * G::WDUMP;
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) pw[((k * NCT + aw) * 4 + r) * 64 + lane] = wacc[k][r];   // only this wave's co-tile (RedArgs::split_a)
    } else {
        // C = 16: the four waves hold the same (co, ci, tap) elements: summed through LDS, ONE dump per workgroup (wave slot 0) --
        // the reduce kernel reads a quarter of the bytes (RedArgs::one_dump)
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave * G::WDUMP + (k * 4 + r) * 64 + lane] = wacc[k][r];
        __syncthreads();
        float* pw = part_w + (long)blockIdx.x * 4 * G::WDUMP;
        for (int i = tid; i < G::WDUMP; i += NT) pw[i] = (red[i] + red[G::WDUMP + i]) + (red[2 * G::WDUMP + i] + red[3 * G::WDUMP + i]);
    }
}

// ---- launchers -------------------------------------------------------------------------------------------------------
template <int C, int D, int MODE, bool SAVE>
int launch_conv(const e16* x, const float* w1, const float* b1, const float* w2, const float* b2, const e16* res,
                e16* y, e16* h1, int B, int H, int T, hipStream_t st, const float* jw = nullptr, int jB = 1) {
    using G = WT<C, D>;
    const int tiles_h = (H + G::TH - 1) / G::TH, tiles_t = (T + G::TW - 1) / G::TW, ntiles = B * tiles_h * tiles_t;
    static const int per_cu = tt_tune("TTRAP_WIDE_PER_CU", 4);
    // (a forward kernel capped at 168 registers for a third workgroup per CU at C = 32 was measured in round 3 and bought nothing: DESIGN.md 7b)
    static AttrOnce once;
    auto kern = k_wrb_conv<C, D, MODE, SAVE, 2>;
    if (int rc = raise_lds(kern, G::LDS_BYTES, once)) return rc;
    // registers admit two workgroups per CU at C = 32 (forward) whatever the LDS would hold: do not launch a third that only runs
    // after the first two have finished (a tail on a third of the chip)
    const int cap = (C == 32 && MODE != 1 && per_cu > 2) ? 2 : per_cu;
    hipLaunchKernelGGL(kern, dim3(grid_for(ntiles, G::LDS_BYTES, cap)), dim3(NT), G::LDS_BYTES, st, x, w1, b1, w2, b2, res, y, h1,
                       B, H, T, tiles_h, tiles_t, ntiles, jw, jB);
    TT_LAUNCH_CHECK();
    return 0;
}

template <int C> constexpr long dump_floats() { return (long)MAX_A_WG * WA<C>::DUMP + (long)MAX_W_WG * 4 * 9 * (C / 16) * 256; }

template <int C, int D, int TH, int TW>
int launch_dxw(const e16* x, const e16* da1, const e16* dy, const float* w1, e16* dx, float* part_w, float* part_a, int grid_a,
               float* dw1, float* db1, float* dw2, float* db2, int B, int H, int T, hipStream_t st) {
    using X = DXW<C, D, TH, TW>;
    const int tiles_h = (H + TH - 1) / TH, tiles_t = (T + TW - 1) / TW, ntiles = B * tiles_h * tiles_t;
    static const int x_per_cu = tt_tune("TTRAP_DXW_PER_CU", C == 32 ? 3 : 4);      // registers: 165 / 113 VGPRs
    int gx = grid_for(ntiles, X::LDS_BYTES, x_per_cu);
    if (gx > MAX_W_WG) gx = MAX_W_WG;
    bool gated = false;
    if constexpr (D == 1) gated = ttx_gate_dx == 1;              // a level's first block: dx leaves gated (k_nrb_bwd_fused, GOUT)
    if (gated) {
        if constexpr (D == 1) {
            if (ttx_skip.g) {
                static AttrOnce once_j;
                auto kj = k_wrb_dxw<C, D, TH, TW, true, true>;
                if (int rc = raise_lds(kj, X::LDS_BYTES, once_j)) return rc;
                hipLaunchKernelGGL(kj, dim3(gx), dim3(NT), X::LDS_BYTES, st, x, da1, dy, w1, dx, part_w, B, H, T, tiles_h, tiles_t, ntiles, ttx_skip);
                ttx_skip.g = nullptr;
            } else {
                static AttrOnce once_g;
                auto kg = k_wrb_dxw<C, D, TH, TW, true>;
                if (int rc = raise_lds(kg, X::LDS_BYTES, once_g)) return rc;
                hipLaunchKernelGGL(kg, dim3(gx), dim3(NT), X::LDS_BYTES, st, x, da1, dy, w1, dx, part_w, B, H, T, tiles_h, tiles_t, ntiles, SkipJ());
            }
            ttx_gate_dx = 2;
        }
    } else {
        static AttrOnce once_x;
        auto kx = k_wrb_dxw<C, D, TH, TW, false>;
        if (int rc = raise_lds(kx, X::LDS_BYTES, once_x)) return rc;
        hipLaunchKernelGGL(kx, dim3(gx), dim3(NT), X::LDS_BYTES, st, x, da1, dy, w1, dx, part_w, B, H, T, tiles_h, tiles_t, ntiles, SkipJ());
    }
    TT_LAUNCH_CHECK();
    RedArgs ra{part_w, gx, part_a, grid_a, dw1, db1, dw2, db2, C == 32 ? 1 : 0, C == 32 ? 0 : 1};
    constexpr int total = 9 * C * C + C * C + 2 * C;
    return reduce_or_defer(k_wrb_reduce<C>, total, ra, st);
}

// One-pass backward (k_wrb_bwds, conv_level_bf16.hip: column strips, dA1 in a rolling LDS ring, 4 tensors of HBM traffic) or the
// per-stage kernels below (7 tensors): TTRAP_WBWD1 = 0 / 1 forces one of them for every width and dilation, 16 / 32 the one-pass
// kernel at that width only; unset = the measured choice.  Round 4, measured INSIDE the 64-clip train step (bench.py --timed-only, same
// box, back to back; profiles/r04_wbwd1_step_ab.txt), average call: C = 16 0.418 -> 0.376 ms (taken: step 57.64 -> 56.75 ms);
// C = 32 0.427 -> 0.495 ms (not taken).  The isolated micro-benchmark (tools/kb_level.py) had the C = 32 strips level with the
// per-stage kernels (0.450 / 0.458 / 0.453 vs 0.465 / 0.430 / 0.464 ms): inside the step the per-stage path finds dy (just written by
// the next block's backward) and its own dA1 in the 256 MB memory-side cache, so its extra passes cost less than their bytes, while
// the one-pass kernel runs at two to three waves per SIMD (168 registers) with the vector ALUs 39 % and the matrix pipe 19 % busy.
inline int onepass_choice(int C, int D) {
    static const int forced = tt_switch("TTRAP_WBWD1", -1);       // 0 / 1: never / always; 16 / 32: at that width only
    if (forced == 16 || forced == 32) return C == forced ? 1 : 0;
    if (forced >= 0) return forced ? 1 : 0;
    (void)D;
    return C == 16 ? 1 : 0;
}

template <int C, int D>
int launch_bwd(const e16* x, const e16* h1, const e16* dy, const float* w1, const float* w2, const float* b2,
               e16* dx, float* dw1, float* db1, float* dw2, float* db2, unsigned char* ws, int B, int H, int T,
               hipStream_t st) {
    if (onepass_choice(C, D)) return tt_wide_rb_bwd_onepass(x, h1, dy, w1, w2, b2, dx, dw1, db1, dw2, db2, ws, B, C, H, T, D, st);
    const long npix = (long)B * H * T;
    e16* da1 = reinterpret_cast<e16*>(ws);
    float* part_a = reinterpret_cast<float*>(ws + ((npix * C * 2 + 255) / 256) * 256);
    float* part_w = part_a + (long)MAX_A_WG * WA<C>::DUMP;
    // pointwise chain
    using G = WA<C>;
    static AttrOnce once;
    auto kern = k_wrb_bwd_a<C>;
    if (int rc = raise_lds(kern, G::LDS_BYTES, once)) return rc;
    const long ngroups = (npix + 15) / 16;
    const long want = (ngroups + 3) / 4;
    static const int a_per_cu = tt_tune("TTRAP_BWDA_PER_CU", 4);
    int grid = (int)(want < (long)a_per_cu * tt_cus() ? want : (long)a_per_cu * tt_cus());
    if (grid > MAX_A_WG) grid = MAX_A_WG;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), G::LDS_BYTES, st, h1, dy, w2, b2, da1, part_a, npix, ngroups);
    TT_LAUNCH_CHECK();
    // data gradient + weight gradient in one pass (TTRAP_DXW=0: the two separate kernels)
    static const int dxw = tt_switch("TTRAP_DXW", 1);
    if (dxw) {
        // tiles 8 x 32 at both widths (round 3, per call at the bench shape, separate kernels -> merged: C = 32 0.486 / 0.485 / 0.521 ->
        // 0.413 / 0.443 / 0.454 ms; C = 16 0.469 / 0.469 / 0.474 -> 0.430 / 0.431 / 0.451 ms; 8 x 64 tiles at C = 16 lose at
        // dilation 2, 3 (0.50 ms: two workgroups per CU); 16 x 32 tiles at C = 16: 0.431 / 0.439 / 0.496 ms)
        // C = 32, dilation 2: two images of 12 x 36 pixels are 55 KB = two workgroups per CU; 6-row tiles (49 KB) admit the third:
        // 0.456 -> 0.427 ms per call.  Dilation 3 (68 KB; 4-row tiles 49 KB): no change, 0.467 ms either way -- 8 rows kept.
        // Round 5: the x tile without halo: 8-row tiles fit three workgroups per CU at every dilation
        // (the round-3 / 4 form -- x with halo, 6-row tiles at C = 32 / dilation 2 -- is gone: profiles/r05_dxw_xh_x3n_shapes_ab.txt)
        return launch_dxw<C, D, 8, 32>(x, da1, dy, w1, dx, part_w, part_a, grid, dw1, db1, dw2, db2, B, H, T, st);
    }
    // data gradient
    if (int rc = launch_conv<C, D, 1, false>(da1, w1, nullptr, nullptr, nullptr, dy, dx, nullptr, B, H, T, st)) return rc;
    // weight gradient
    using W = WG<C, D>;
    static AttrOnce once_w;
    auto kw = k_wrb_wgrad<C, D>;
    if (int rc = raise_lds(kw, W::LDS_BYTES, once_w)) return rc;
    const int tiles_h = (H + W::TH - 1) / W::TH, tiles_t = (T + W::TW - 1) / W::TW, ntiles = B * tiles_h * tiles_t;
    static const int w_per_cu = tt_tune("TTRAP_WGRAD_PER_CU", 2);
    int gw = grid_for(ntiles, W::LDS_BYTES, w_per_cu);
    if (gw > MAX_W_WG) gw = MAX_W_WG;
    hipLaunchKernelGGL(kw, dim3(gw), dim3(NT), W::LDS_BYTES, st, x, da1, part_w, B, H, T, tiles_h, tiles_t, ntiles);
    TT_LAUNCH_CHECK();
    RedArgs ra{part_w, gw, part_a, grid, dw1, db1, dw2, db2};
    constexpr int total = 9 * C * C + C * C + 2 * C;
    return reduce_or_defer(k_wrb_reduce<C>, total, ra, st);
}

template <int C>
int fwd_c(const e16* x, const float* w1, const float* b1, const float* w2, const float* b2, e16* y, e16* h1, int B,
          int H, int T, int D, hipStream_t st, const e16* je = nullptr, const float* jw = nullptr, int jB = 1) {
#define TT_WFWD(DD)                                                                                                  \
    if (je) return h1 ? launch_conv<C, DD, 2, true>(x, w1, b1, w2, b2, je, y, h1, B, H, T, st, jw, jB)                \
                      : launch_conv<C, DD, 2, false>(x, w1, b1, w2, b2, je, y, nullptr, B, H, T, st, jw, jB);         \
    return h1 ? launch_conv<C, DD, 0, true>(x, w1, b1, w2, b2, nullptr, y, h1, B, H, T, st)                           \
              : launch_conv<C, DD, 0, false>(x, w1, b1, w2, b2, nullptr, y, nullptr, B, H, T, st)
    switch (D) {
        case 1: TT_WFWD(1);
        case 2: TT_WFWD(2);
        case 3: TT_WFWD(3);
    }
#undef TT_WFWD
    return TT_E_UNSUPPORTED;
}
template <int C>
int bwd_c(const e16* x, const e16* h1, const e16* dy, const float* w1, const float* w2, const float* b2, e16* dx,
          float* dw1, float* db1, float* dw2, float* db2, unsigned char* ws, int B, int H, int T, int D, hipStream_t st) {
    switch (D) {
        case 1: return launch_bwd<C, 1>(x, h1, dy, w1, w2, b2, dx, dw1, db1, dw2, db2, ws, B, H, T, st);
        case 2: return launch_bwd<C, 2>(x, h1, dy, w1, w2, b2, dx, dw1, db1, dw2, db2, ws, B, H, T, st);
        case 3: return launch_bwd<C, 3>(x, h1, dy, w1, w2, b2, dx, dw1, db1, dw2, db2, ws, B, H, T, st);
    }
    return TT_E_UNSUPPORTED;
}

// ==== narrow levels (C = 4, 8) ========================================================================================
// A pixel is 8 / 16 bytes, so here a LANE is a pixel: v_mfma_f32_4x4x4_16b_bf16 runs sixteen independent 4x4x4 products
// per wave -- block = lane / 4, A row / B column = lane % 4, each lane's four bf16 are k = 0..3, D register = row -- so with
// A = a 4 x 4 slice of the weights (the same in every block) and B = four channels of the lane's own pixel, every lane
// ends up with ALL output channels of its pixel: a 64-pixel row segment per wave instruction, 8- / 16-byte accesses
// everywhere, nothing wasted on padding the channel count up to a 16-row tile.

template <int C> struct VecOf { typedef typename std::conditional<C == 8, e16x8, e16x4>::type type; };
template <int C> __device__ __forceinline__ s16x4 chunk_of(const typename VecOf<C>::type& v, int kb) {
    if constexpr (C == 8) {
        const e16x4 h = kb == 0 ? __builtin_shufflevector(v, v, 0, 1, 2, 3) : __builtin_shufflevector(v, v, 4, 5, 6, 7);
        return __builtin_bit_cast(s16x4, h);
    } else {
        return __builtin_bit_cast(s16x4, v);
    }
}

template <int C, int D> struct NTl {
    static constexpr int TH = 16, TW = 64;
    static constexpr int PXB = C * 2;                            // bytes per pixel
    static constexpr int PPP = 16 / PXB;                         // pixels per 16-byte piece: 2 (C = 4), 1 (C = 8)
    static constexpr int DP = (D + PPP - 1) / PPP * PPP;         // column halo in whole pieces (T must be a multiple of PPP)
    static constexpr int RW = TW + 2 * DP, ROWS = TH + 2 * D;
    static constexpr int NP = ROWS * RW / PPP;
    static constexpr int NPR = (NP + NT - 1) / NT * NT;
    static constexpr int LDS_BYTES = NPR * 16;
};

// C = 8 forward: capped at 168 VGPRs (three waves per SIMD; 208 otherwise): 0.194 / 0.190 / 0.195 -> 0.187 / 0.186 / 0.189 ms (library A/B, round 3)
template <int C, int D, int MODE, bool SAVE>
__global__ __launch_bounds__(NT, (C == 8 && MODE != 1) ? 3 : 1) void k_nrb_conv(const e16* __restrict__ x, const float* __restrict__ w1,
                                                 const float* __restrict__ b1, const float* __restrict__ w2,
                                                 const float* __restrict__ b2, const e16* __restrict__ res,
                                                 e16* __restrict__ y, e16* __restrict__ h1, int B, int H, int T,
                                                 int tiles_h, int tiles_t, int ntiles, const float* __restrict__ jw, int jB) {
    using G = NTl<C, D>;
    constexpr bool FWD = MODE != 1, JOIN = MODE == 2;            // MODE 2: the forward with a skip join in its epilogue (see k_wrb_conv)
    float jsc = 0.f;
    if constexpr (JOIN) jsc = jw ? jw[0] : 1.f;
    typedef typename VecOf<C>::type vec_t;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i4 = lane & 3;
    constexpr int NB = C / 4;

    s16x4 A[9][NB][NB], A2[NB][NB];
    float chk = 0.f;                                             // forward: non-finite parameters (poison_acc, bf16_common.h)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ob = 0; ob < NB; ++ob)
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {
                e16x4 t;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int mo = 4 * ob + i4, kc = 4 * kb + k;
                    const float wv = FWD ? w1[(mo * C + kc) * 9 + tap] : w1[(kc * C + mo) * 9 + (8 - tap)];
                    if (FWD) chk = poison_acc(chk, wv);
                    t[k] = (e16)wv;
                }
                A[tap][ob][kb] = __builtin_bit_cast(s16x4, t);
            }
    float b1r[C], b2r[C];
    if constexpr (FWD) {
#pragma unroll
        for (int ob = 0; ob < NB; ++ob)
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {
                e16x4 t;
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float wv = w2[(4 * ob + i4) * C + 4 * kb + k]; chk = poison_acc(chk, wv); t[k] = (e16)wv; }
                A2[ob][kb] = __builtin_bit_cast(s16x4, t);
            }
#pragma unroll
        for (int c = 0; c < C; ++c) { b1r[c] = b1[c]; b2r[c] = b2[c]; chk = poison_acc(poison_acc(chk, b1r[c]), b2r[c]); }
    }
    const bool poisoned = FWD && params_poisoned(chk);          // see k_wrb_conv (a non-finite join weight propagates through the fma by itself)

    const e16* zero = reinterpret_cast<const e16*>(&g_wzero16);
    for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
        int tile = xcd_order(v, ntiles);
        int half = 0;                                            // JOIN over two batches: the halves' tiles side by side in the walk (see k_wrb_conv)
        if (JOIN && B == 2 * jB) { half = tile & 1; tile >>= 1; }
        const int tt = tile % tiles_t; tile /= tiles_t;
        const int th = tile % tiles_h;
        const int b = tile / tiles_h + half * jB, h0 = th * G::TH, t0 = tt * G::TW;
        const e16* xb = x + (long)b * H * T * C;

        __syncthreads();                                         // the previous tile has been consumed
        // (a border-free fast path with per-thread precomputed offsets, as in k_nrb_bwd_fused, bought nothing here: 53.42 vs 53.47 ms per
        // step, profiles/r05_barrier_fastp_ab.txt -- the forward is not bound by its staging arithmetic)
        for (int i = wave * 64; i < G::NPR; i += NT) {
            const int p = i + lane, q = p * G::PPP;
            const int row = q / G::RW, px = q - row * G::RW;
            const int h = h0 - D + row, t = t0 - G::DP + px;
            const bool ok = p < G::NP && (unsigned)h < (unsigned)H && (unsigned)t < (unsigned)T;
            glds16(ok ? xb + ((long)h * T + t) * C : zero, smem + (long)i * 16);
        }
        const int t = t0 + lane;
        const bool valid = t < T;
        vec_t jq;                                                // JOIN: the embedding's pixels one row of this wave ahead (see k_wrb_conv)
        const e16* jrow = nullptr;
        if constexpr (JOIN) {
            jrow = res + (((long)(b % jB) * H) * T + (valid ? t : T - 1)) * C;
            jq = *reinterpret_cast<const vec_t*>(jrow + (long)(h0 + wave < H ? h0 + wave : H - 1) * T * C);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        for (int r = wave; r < G::TH; r += 4) {
            const int h = h0 + r;
            if (h >= H) break;
            const long pix = ((long)b * H + h) * T + t;
            if (poisoned) {
                if (valid) {
                    vec_t qn;
#pragma unroll
                    for (int c = 0; c < C; ++c) qn[c] = (e16)__builtin_nanf("");
                    *reinterpret_cast<vec_t*>(y + pix * C) = qn;
                    if (SAVE) *reinterpret_cast<vec_t*>(h1 + pix * C) = qn;
                }
                continue;
            }
            vec_t rq;
            if constexpr (MODE == 1) rq = *reinterpret_cast<const vec_t*>(res + (valid ? pix : pix - (t - (T - 1))) * C);
            f32x4 acc[NB];
#pragma unroll
            for (int ob = 0; ob < NB; ++ob) {
                if constexpr (FWD) acc[ob] = f32x4{b1r[4 * ob], b1r[4 * ob + 1], b1r[4 * ob + 2], b1r[4 * ob + 3]};
                else acc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            vec_t centre;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kh = tap / 3, kw = tap - 3 * kh;
                const int pxi = (r + kh * D) * G::RW + G::DP + lane + (kw - 1) * D;
                const vec_t bq = *reinterpret_cast<const vec_t*>(smem + (long)pxi * G::PXB);
                if (tap == 4) centre = bq;
#pragma unroll
                for (int ob = 0; ob < NB; ++ob)
#pragma unroll
                    for (int kb = 0; kb < NB; ++kb) acc[ob] = mma4(A[tap][ob][kb], chunk_of<C>(bq, kb), acc[ob]);
            }
            if constexpr (JOIN) {
                rq = jq;
                const int hn = h + 4 < H ? h + 4 : H - 1;
                jq = *reinterpret_cast<const vec_t*>(jrow + (long)hn * T * C);
            }
            vec_t o;
            if constexpr (MODE == 1) {
#pragma unroll
                for (int c = 0; c < C; ++c) o[c] = (e16)(acc[c >> 2][c & 3] + (float)rq[c]);
            } else {
                vec_t hq;
#pragma unroll
                for (int c = 0; c < C; ++c) hq[c] = (e16)elu_res(acc[c >> 2][c & 3]);
                if (SAVE && valid) h1_store(reinterpret_cast<vec_t*>(h1 + pix * C), hq);
                f32x4 z[NB];
#pragma unroll
                for (int ob = 0; ob < NB; ++ob) {
                    z[ob] = f32x4{b2r[4 * ob], b2r[4 * ob + 1], b2r[4 * ob + 2], b2r[4 * ob + 3]};
#pragma unroll
                    for (int kb = 0; kb < NB; ++kb) z[ob] = mma4(A2[ob][kb], chunk_of<C>(hq, kb), z[ob]);
                }
#pragma unroll
                for (int c = 0; c < C; ++c)
                    o[c] = JOIN ? (e16)__builtin_fmaf(jsc, (float)rq[c], elu_out(z[c >> 2][c & 3]) + (float)centre[c]) : (e16)(elu_out(z[c >> 2][c & 3]) + (float)centre[c]);
            }
            if (valid) *reinterpret_cast<vec_t*>(y + pix * C) = o;
        }
    }
}

// Pointwise chain of the backward, lane = pixel.  dW2 is a per-lane outer product (C^2 fused multiply-adds per pixel),
// reduced over the wave by shuffles once at the end.  One dump [dW2 co*C+ci][db1][db2] per workgroup.
template <int C>
__global__ __launch_bounds__(NT) void k_nrb_bwd_a(const e16* __restrict__ h1, const e16* __restrict__ dy,
                                                  const float* __restrict__ w2, const float* __restrict__ b2,
                                                  e16* __restrict__ da1, float* __restrict__ part, long npix, long ngroups) {
    typedef typename VecOf<C>::type vec_t;
    constexpr int NB = C / 4, DUMP = C * C + 2 * C;
    __shared__ float red[4][DUMP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i4 = lane & 3;
    s16x4 A2[NB][NB], A2T[NB][NB];
#pragma unroll
    for (int ob = 0; ob < NB; ++ob)
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
            e16x4 a, at;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a[k] = (e16)w2[(4 * ob + i4) * C + 4 * kb + k];
                at[k] = (e16)w2[(4 * kb + k) * C + 4 * ob + i4];
            }
            A2[ob][kb] = __builtin_bit_cast(s16x4, a); A2T[ob][kb] = __builtin_bit_cast(s16x4, at);
        }
    float b2r[C], acc[DUMP];
#pragma unroll
    for (int c = 0; c < C; ++c) b2r[c] = b2[c];
#pragma unroll
    for (int e = 0; e < DUMP; ++e) acc[e] = 0.f;

    vec_t zero_v;
#pragma unroll
    for (int c = 0; c < C; ++c) zero_v[c] = (e16)0.f;
    const long gstride = (long)gridDim.x * 4;
    long grp = (long)blockIdx.x * 4 + wave;
    vec_t hq = zero_v, dq = zero_v;
    if (grp < ngroups && grp * 64 + lane < npix) {
        hq = *reinterpret_cast<const vec_t*>(h1 + (grp * 64 + lane) * C);
        dq = *reinterpret_cast<const vec_t*>(dy + (grp * 64 + lane) * C);
    }
    for (; grp < ngroups; grp += gstride) {
        const long pix = grp * 64 + lane, pn = (grp + gstride) * 64 + lane;
        vec_t hq_n = zero_v, dq_n = zero_v;
        if (grp + gstride < ngroups && pn < npix) {
            hq_n = *reinterpret_cast<const vec_t*>(h1 + pn * C);
            dq_n = *reinterpret_cast<const vec_t*>(dy + pn * C);
        }
        f32x4 z[NB], u[NB];
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) {
            z[ob] = f32x4{b2r[4 * ob], b2r[4 * ob + 1], b2r[4 * ob + 2], b2r[4 * ob + 3]};
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) z[ob] = mma4(A2[ob][kb], chunk_of<C>(hq, kb), z[ob]);
        }
        float gv[C], hv[C], gr[C];
        vec_t gq, aq;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float a2 = z[c >> 2][c & 3];
            gv[c] = (float)dq[c] * elu_dpre(a2);
            gq[c] = (e16)gv[c];
            gr[c] = (float)gq[c];
            hv[c] = (float)hq[c];
        }
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) {
            u[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) u[ob] = mma4(A2T[ob][kb], chunk_of<C>(gq, kb), u[ob]);
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float a1 = u[c >> 2][c & 3] * elu_dout(hv[c]);
            aq[c] = (e16)a1;
            acc[C * C + c] += a1; acc[C * C + C + c] += gv[c];       // invalid pixels contribute zeros
        }
        if (pix < npix) *reinterpret_cast<vec_t*>(da1 + pix * C) = aq;
#pragma unroll
        for (int co = 0; co < C; ++co)
#pragma unroll
            for (int ci = 0; ci < C; ++ci)     // plain v_fmac_f32, spelled out (the SLP-vectorised form of this pattern was the run-to-run different one of round 3: profiles/r04_isa_*)
                asm("v_fmac_f32 %0, %1, %2" : "+v"(acc[co * C + ci]) : "v"(gr[co]), "v"(hv[ci]));
        hq = hq_n; dq = dq_n;
    }
#pragma unroll
    for (int e = 0; e < DUMP; ++e) {
        float sv = acc[e];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sv += __shfl_xor(sv, o, 64);
        if (lane == 0) red[wave][e] = sv;
    }
    __syncthreads();
    float* pw = part + (long)blockIdx.x * DUMP;
    for (int e = tid; e < DUMP; e += NT) pw[e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
}

// 3x3 weight gradient of the narrow levels.  A row of the channel-innermost image is read as a flat run of 32-byte slots
// (16 bf16 = 2 pixels at C = 8, 4 pixels at C = 4); the transpose reads then deliver, per lane, "column" (pixel-in-slot s,
// channel c) = 4 s.. over the slots that are the K of the product.  D[(s,c)][(s',c')] mixes the pixels of a slot; only the
// diagonal blocks s = s' pair a gradient pixel with its own input pixel -- k_nrb_reduce adds exactly those.
template <int C, int D> struct NG {
    static constexpr int TH = 8, TW = 128;
    static constexpr int PXB = C * 2, PPP = 16 / PXB;
    static constexpr int DP = (D + PPP - 1) / PPP * PPP;
    static constexpr int RW = TW + 2 * DP, ROWS = TH + 2 * D;
    static constexpr int XP = ROWS * RW / PPP, GP = TH * TW / PPP;
    static constexpr int XPR = (XP + NT - 1) / NT * NT, GPR = (GP + NT - 1) / NT * NT;
    static constexpr int X_BYTES = XPR * 16, G_BYTES = GPR * 16;
    static constexpr int CHUNKS = TW * PXB / 1024;               // 1024-byte chunks (32 slots) per row: 1 (C = 4), 2 (C = 8)
    static constexpr int DUMP = 9 * 256;
    static constexpr int LDS_BYTES = X_BYTES + G_BYTES;
};

template <int C, int D>
__global__ __launch_bounds__(NT) void k_nrb_wgrad(const e16* __restrict__ x, const e16* __restrict__ da1,
                                                  float* __restrict__ part, int B, int H, int T, int tiles_h, int tiles_t,
                                                  int ntiles) {
    using G = NG<C, D>;
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* xs = smem;
    unsigned char* gs = smem + G::X_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int trj = n >> 2, trq = n & 3;
    const int chunk = wave % G::CHUNKS, row0 = wave / G::CHUNKS;
    constexpr int RSTEP = 4 / G::CHUNKS;
    f32x4 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};

    const e16* zero = reinterpret_cast<const e16*>(&g_wzero16);
    for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
        int tile = xcd_order(v, ntiles);
        const int tt = tile % tiles_t; tile /= tiles_t;
        const int th = tile % tiles_h;
        const int b = tile / tiles_h, h0 = th * G::TH, t0 = tt * G::TW;
        const e16* xb = x + (long)b * H * T * C;
        const e16* gb = da1 + (long)b * H * T * C;
        __syncthreads();
        for (int i = wave * 64; i < G::XPR; i += NT) {
            const int p = i + lane, q = p * G::PPP;
            const int row = q / G::RW, px = q - row * G::RW;
            const int h = h0 - D + row, t = t0 - G::DP + px;
            const bool ok = p < G::XP && (unsigned)h < (unsigned)H && (unsigned)t < (unsigned)T;
            glds16(ok ? xb + ((long)h * T + t) * C : zero, xs + (long)i * 16);
        }
        for (int i = wave * 64; i < G::GPR; i += NT) {
            const int p = i + lane, q = p * G::PPP;
            const int row = q / G::TW, px = q - row * G::TW;
            const int h = h0 + row, t = t0 + px;
            const bool ok = p < G::GP && h < H && t < T;
            glds16(ok ? gb + ((long)h * T + t) * C : zero, gs + (long)i * 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        const int so = chunk * 1024 + 32 * (4 * g + trj) + 8 * trq;      // byte offset inside a row of this lane's first slot
        for (int r = row0; r < G::TH; r += RSTEP) {
            if (h0 + r >= H) break;
            const unsigned char* gp = gs + (long)r * (G::TW * G::PXB) + so;
            const s16x4 glo = lds_tr16(gp), ghi = lds_tr16(gp + 512);
            const e16x8 ga = __builtin_bit_cast(e16x8, __builtin_shufflevector(glo, ghi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int kh = k / 3, kw = k - 3 * kh;
                const unsigned char* xp = xs + (long)(r + kh * D) * (G::RW * G::PXB) + (G::DP + (kw - 1) * D) * G::PXB + so;
                const s16x4 lo = lds_tr16(xp), hi = lds_tr16(xp + 512);
                acc[k] = mma32(ga, __builtin_bit_cast(e16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7)), acc[k]);
            }
        }
    }
    float* pw = part + ((long)blockIdx.x * 4 + wave) * G::DUMP;
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) pw[(k * 4 + r) * 64 + lane] = acc[k][r];
}

// ---- the whole backward of a narrow block in ONE pass ---------------------------------------------------------------------
// h1 and dy tiles (16 x 64 pixels + halo) and the x tile arrive by LDS-DMA; phase 1 runs the pointwise chain on EVERY pixel of the
// halo'd tile (a lane = a pixel) and writes dA1 over h1 in LDS -- out-of-image pixels have h1 = dy = 0 and give dA1 = 0, which is the
// zero padding of the data gradient; db1 / db2 / dW2 are accumulated over the tile's own (centre, in-image) pixels only -- dW2 as ONE
// K = pixels matrix product per 64 pixels (the 32-byte slots of the flat images are the operand rows; the masked dA2 goes through a
// 64-pixel buffer per wave, both operands by transpose reads, the diagonal blocks of the 16 x 16 accumulator are the gradient).
// Phase 2 takes dx = dy + W1^T (*) dA1 and dW1 straight from the LDS images.  dA1 never goes to HBM: 4 tensors of traffic per block
// instead of 8 (h1, dy, x read; dx written), one launch instead of three.  The A operands of every phase sit in a 16-bit weight
// image in LDS built once per workgroup: entry ((m * NB + ob) * NB + kb) * 4 + i4 holds the four values lane i4 of a 4-lane group feeds
// to product (ob, kb) of matrix m (0: W2, 1: W2^T, 2 + tap: W1^T of tap 8 - tap).
// Round 5 (against the round-4 cut of this kernel, A/B in profiles/r05_nbf2_ab.txt: 55.4 -> 54.5 ms per step; C = 8 / dilation 3 0.457 ->
// 0.323 ms per call, where the old cut had to fall back to the per-stage kernels), re-cut where its counters said the time goes:
// (i)   the x tile is staged WITHOUT halo: dW1[tap] = sum_q x[q] (x) dA1[q - off(tap)] indexes the weight gradient by the pixel of x
//       (the strip kernel's identity, conv_level_bf16.hip), and dA1 has its halo anyway -- a third of the LDS image bytes and DMA
//       instructions less, and C = 8 / dilation 3 fits two workgroups per CU (86 -> 78 KB), which the old cut did not;
// (ii)  x is not needed before phase 2b: its DMA is issued last and waited for only behind phase 1 (s_waitcnt vmcnt(NITX) in front);
// (iii) phase 1's per-step index arithmetic (flat pixel -> row / column by division, six compares for "this tile's own pixel") is
//       tile-independent: LDS offset and a packed (row, column) key are computed once per kernel, the tile contributes two scalars;
// (iv)  one conversion of dA2 to 16 bits serves the matrix operand and the dW2 staging buffer (the compiler emitted both forms).
// (Round 5, measured and dropped at C = 8: three workgroups per CU instead of two -- data-gradient weights read per tap from the LDS image instead
// of held in 72 registers (142-157 registers), tiles of 12 / 12 / 8 rows so that three sets of images fit: 51.59 / 51.39 ms per step against
// 51.64 / 51.21; 8-row tiles throughout: 52.16 / 51.94.  profiles/r05_nbf2_c8_occupancy_ab.txt)
#ifndef TT_NBF2_MINW4
#define TT_NBF2_MINW4 4          // C = 4: registers capped for four workgroups per CU (the LDS limit); 1 lets the compiler take 156 (three)
#endif
// GOUT: the block is the first of its level and the layer in front of it ends in an ELU whose output IS this block's x (the transposed /
// strided layer of a DecoderBlock / EncoderBlock): dx leaves as dx * ELU'(x) -- the gated gradient that layer's backward needs, so that it
// reads neither its saved output nor stages anything through registers (tt_wide_level_bwd_gated, conv_stride_bf16.hip).
// SJ (with GOUT): the backward of a skip join on this block's input rides on the gated epilogue (SkipJ, wide_common.h)
template <int C, int D, bool GOUT, bool SJ = false>
__global__ __launch_bounds__(NT, C == 8 ? 2 : TT_NBF2_MINW4) void k_nrb_bwd_fused(const e16* __restrict__ x, const e16* __restrict__ h1,
                                                       const e16* __restrict__ dy, const float* __restrict__ w1,
                                                       const float* __restrict__ w2, const float* __restrict__ b2,
                                                       e16* __restrict__ dx, float* __restrict__ part_a, float* __restrict__ part_w,
                                                       int B, int H, int T, int tiles_h, int tiles_t, int ntiles, SkipJ sj = SkipJ()) {
    static_assert(!SJ || GOUT, "a skip join rides on the gated epilogue only");
    using G = NTl<C, D>;
    typedef typename VecOf<C>::type vec_t;
    float sjw = 0.f, sjdot = 0.f;
    if constexpr (SJ) sjw = sj.w ? sj.w[0] : 1.f;
    // the halo'd images hold exactly their NP pieces (the DMA instructions' tail lanes are masked off, not written as zeros behind the
    // image): at C = 4 / dilation 3 the whole-instruction rounding (770 -> 1024 pieces, twice) was the difference between three and
    // four workgroups per CU
    constexpr int NB = C / 4, ADUMP = C * C + 2 * C, IMG = (G::NP * 16 + 63) / 64 * 64, NPX = G::ROWS * G::RW;
    constexpr int XIMG = G::TH * G::TW * G::PXB, NITX = XIMG / 16 / NT, XROWB = G::TW * G::PXB;
    constexpr int NS1 = (NPX + NT - 1) / NT;                     // phase-1 steps of 64 pixels per wave
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* hs = smem;                                    // h1, then dA1   (halo'd)
    unsigned char* gs = smem + IMG;                              // dy             (halo'd)
    unsigned char* xs = smem + 2 * IMG;                          // x              (the tile's own pixels)
    unsigned char* wimg = xs + XIMG;                             // A operands of every phase, 16-bit
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i4 = lane & 3;
    const int n = lane & 15, g = lane >> 4, trj = n >> 2, trq = n & 3;

    float b2r[C], acc[2 * C];                                    // db1 | db2
#pragma unroll
    for (int c = 0; c < C; ++c) b2r[c] = b2[c];
#pragma unroll
    for (int e = 0; e < 2 * C; ++e) acc[e] = 0.f;
    f32x4 wacc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wacc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 dw2m = f32x4{0.f, 0.f, 0.f, 0.f};
    const e16* zero = reinterpret_cast<const e16*>(&g_wzero16);
    unsigned char* abuf = wimg + 11 * NB * NB * 32 + wave * (64 * G::PXB);      // this wave's 64 pixels of masked dA2
    vec_t zero_px;
#pragma unroll
    for (int c = 0; c < C; ++c) zero_px[c] = (e16)0.f;
    for (int e = tid; e < (2 + 9) * NB * NB * 4; e += NT) {
        const int l4 = e & 3, kb = (e >> 2) % NB, ob = (e >> 2) / NB % NB, m = (e >> 2) / (NB * NB);
        e16x4 a;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            a[k] = (e16)(m == 0 ? w2[(4 * ob + l4) * C + 4 * kb + k] : m == 1 ? w2[(4 * kb + k) * C + 4 * ob + l4]
                                   : w1[((4 * kb + k) * C + 4 * ob + l4) * 9 + (8 - (m - 2))]);
        *reinterpret_cast<e16x4*>(wimg + e * 8) = a;
    }
    constexpr int NITD = G::NPR / NT;                            // 16-byte pieces per thread and halo'd image
    unsigned rel[NITD], relx[NITX];                              // byte offsets from the tile's first halo pixel / first own pixel
#pragma unroll
    for (int it = 0; it < NITD; ++it) {
        const int p = wave * 64 + it * NT + lane, q = (p < G::NP ? p : 0) * G::PPP;
        const int row = q / G::RW, px = q - row * G::RW;
        rel[it] = (unsigned)(row * T + px) * (unsigned)(C * 2);
    }
#pragma unroll
    for (int it = 0; it < NITX; ++it) {
        const int q = (wave * 64 + it * NT + lane) * G::PPP;
        const int row = q / G::TW, px = q - row * G::TW;
        relx[it] = (unsigned)(row * T + px) * (unsigned)(C * 2);
    }
    // phase 1, per step s: LDS offset of the lane's pixel (clamped into the image) and key = row | column << 16 of a pixel of the
    // tile proper, ~0 for a halo pixel: "one of this tile's own in-image pixels" is then (key & 0xffff) < hlim && (key >> 16) < tlim
    unsigned poff[NS1], pkey[NS1];
#pragma unroll
    for (int s1 = 0; s1 < NS1; ++s1) {
        const int q = wave * 64 + s1 * NT + lane, qq = q < NPX ? q : NPX - 1;
        const int row = qq / G::RW, col = qq - row * G::RW;
        poff[s1] = (unsigned)qq * G::PXB;
        pkey[s1] = (q < NPX && row >= D && row < D + G::TH && col >= G::DP && col < G::DP + G::TW) ? (unsigned)row | ((unsigned)col << 16) : 0xffffffffu;
    }
    auto wfrag = [&](int m, int ob, int kb) { return *reinterpret_cast<const s16x4*>(wimg + ((((m * NB + ob) * NB + kb) << 2) + i4) * 8); };
    for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
        int tile = xcd_order(v, ntiles);
        const int tt = tile % tiles_t; tile /= tiles_t;
        const int th = tile % tiles_h;
        const int b = tile / tiles_h, h0 = th * G::TH, t0 = tt * G::TW;
        const long ib = (long)b * H * T * C;
        // every wave is done READING the three images of the previous tile (its dx stores may still be in flight: they touch no LDS)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const bool fast = h0 >= D && h0 + G::TH + D <= H && t0 >= G::DP && t0 + G::TW + G::DP <= T;
        if (fast) {
            const long first = ib + ((long)(h0 - D) * T + (t0 - G::DP)) * C;
            const char* hb = reinterpret_cast<const char*>(h1 + first);
            const char* gb = reinterpret_cast<const char*>(dy + first);
            const char* xb = reinterpret_cast<const char*>(x + ib + ((long)h0 * T + t0) * C);
#pragma unroll
            for (int it = 0; it < NITD; ++it) {
                const int i = wave * 64 + it * NT;
                if (it + 1 < NITD || i + lane < G::NP) {         // (only the last round has lanes past the image)
                    glds16_h1(hb + rel[it], hs + (long)i * 16);
                    glds16(gb + rel[it], gs + (long)i * 16);
                }
            }
#pragma unroll
            for (int it = 0; it < NITX; ++it) glds16_x(xb + relx[it], xs + (long)(wave * 64 + it * NT) * 16);
        } else {
            for (int i = wave * 64; i < G::NPR; i += NT) {
                const int p = i + lane, q = p * G::PPP;
                const int row = q / G::RW, px = q - row * G::RW;
                const int h = h0 - D + row, t = t0 - G::DP + px;
                const bool ok = (unsigned)h < (unsigned)H && (unsigned)t < (unsigned)T;
                const long off = ib + ((long)h * T + t) * C;
                if (p < G::NP) {
                    glds16_h1(ok ? h1 + off : zero, hs + (long)i * 16);
                    glds16(ok ? dy + off : zero, gs + (long)i * 16);
                }
            }
#pragma unroll
            for (int it = 0; it < NITX; ++it) {
                const int q = (wave * 64 + it * NT + lane) * G::PPP;
                const int row = q / G::TW, px = q - row * G::TW;
                const bool ok = h0 + row < H && t0 + px < T;
                glds16_x(ok ? x + ib + ((long)(h0 + row) * T + t0 + px) * C : zero, xs + (long)(wave * 64 + it * NT) * 16);
            }
        }
        // h1 and dy complete (this wave's pieces; the barrier collects the others'), the NITX pieces of x still in flight.  A bare
        // s_barrier: __syncthreads() carries a workgroup fence that the compiler lowers to s_waitcnt vmcnt(0) first, which would wait
        // for x after all (LDS-DMA stays in flight across s_barrier; what orders an LDS read behind a DMA piece is the ISSUING wave's
        // vmcnt plus a barrier for the other waves -- /opt/skills/guides/MI355X_MICROARCH.md, "Two waves per SIMD", item 7)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NITX) : "memory");

        // ---- phase 1: pointwise chain on every pixel of the halo'd image; dA1 over h1 ----
        {
            s16x4 A2[NB][NB], A2T[NB][NB];
#pragma unroll
            for (int ob = 0; ob < NB; ++ob)
#pragma unroll
                for (int kb = 0; kb < NB; ++kb) { A2[ob][kb] = wfrag(0, ob, kb); A2T[ob][kb] = wfrag(1, ob, kb); }
            const unsigned hlim = (unsigned)(H - h0 + D), tlim = (unsigned)(T - t0 + G::DP);
#pragma unroll
            for (int s1 = 0; s1 < NS1; ++s1) {
                const int c0 = wave * 64 + s1 * NT;
                if (c0 >= NPX) break;
                const bool centre = (pkey[s1] & 0xffffu) < hlim && (pkey[s1] >> 16) < tlim;
                const vec_t hq = *reinterpret_cast<const vec_t*>(hs + poff[s1]);
                const vec_t dq = *reinterpret_cast<const vec_t*>(gs + poff[s1]);
                f32x4 z[NB], u[NB], gv[NB];
                e16x4 gq4[NB];
#pragma unroll
                for (int ob = 0; ob < NB; ++ob) {
                    z[ob] = f32x4{b2r[4 * ob], b2r[4 * ob + 1], b2r[4 * ob + 2], b2r[4 * ob + 3]};
#pragma unroll
                    for (int kb = 0; kb < NB; ++kb) z[ob] = mma4(A2[ob][kb], chunk_of<C>(hq, kb), z[ob]);
                }
#pragma unroll
                for (int ob = 0; ob < NB; ++ob) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) gv[ob][r] = (float)dq[4 * ob + r] * elu_dpre(z[ob][r]);
                    gq4[ob] = __builtin_convertvector(gv[ob], e16x4);
                }
#pragma unroll
                for (int ob = 0; ob < NB; ++ob) {
                    u[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kb = 0; kb < NB; ++kb) u[ob] = mma4(A2T[ob][kb], __builtin_bit_cast(s16x4, gq4[kb]), u[ob]);
                }
                const float m = centre ? 1.f : 0.f;              // sums only over this tile's own pixels
                vec_t aq, gq;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float a1 = u[c >> 2][c & 3] * elu_dout((float)hq[c]);
                    aq[c] = (e16)a1;
                    gq[c] = gq4[c >> 2][c & 3];
                    acc[c] = __builtin_fmaf(m, a1, acc[c]);
                    acc[C + c] = __builtin_fmaf(m, gv[c >> 2][c & 3], acc[C + c]);
                }
                // dW2 += dA2 (x) h1 over the tile's own pixels as ONE matrix product per 64 pixels
                *reinterpret_cast<vec_t*>(abuf + lane * G::PXB) = centre ? gq : zero_px;
                __builtin_amdgcn_wave_barrier();
                asm volatile("" ::: "memory");
                const unsigned char* hb0 = hs + (long)c0 * G::PXB;                   // the 64 pixels of this step, before dA1 replaces them
                const int so = 32 * (4 * g + trj) + 8 * trq;
                // transpose reads spelled out: in front of the builtin form the compiler places s_waitcnt vmcnt(0) (an LDS read it cannot
                // prove disjoint from the x pieces still arriving by LDS-DMA), which is exactly the wait this kernel defers
                const unsigned ga_ = lds_addr(abuf + so), ha_ = lds_addr(hb0 + so);
                if constexpr (C == 8) {
                    s16x4 glo, ghi, hlo, hhi;
                    asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:512\n\tds_read_b64_tr_b16 %2, %5\n\t"
                                 "ds_read_b64_tr_b16 %3, %5 offset:512\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(glo), "=&v"(ghi), "=&v"(hlo), "=&v"(hhi) : "v"(ga_), "v"(ha_) : "memory");
                    dw2m = mma32(__builtin_bit_cast(e16x8, __builtin_shufflevector(glo, ghi, 0, 1, 2, 3, 4, 5, 6, 7)),
                                 __builtin_bit_cast(e16x8, __builtin_shufflevector(hlo, hhi, 0, 1, 2, 3, 4, 5, 6, 7)), dw2m);
                } else {
                    s16x4 gt_, ht_;
                    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(gt_), "=&v"(ht_) : "v"(ga_), "v"(ha_) : "memory");
                    dw2m = mma16(gt_, ht_, dw2m);
                }
                if (c0 + 64 <= NPX || lane < NPX - c0) *reinterpret_cast<vec_t*>(hs + poff[s1]) = aq;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // x has had all of phase 1 to arrive
        __syncthreads();

        // ---- phase 2a: dx = dy + W1^T (*) dA1 (k_nrb_conv, MODE 1, operands from LDS) ----
        {
            s16x4 A[9][NB][NB];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int ob = 0; ob < NB; ++ob)
#pragma unroll
                    for (int kb = 0; kb < NB; ++kb) A[tap][ob][kb] = wfrag(2 + tap, ob, kb);
            const int t = t0 + lane;
            for (int r = wave; r < G::TH; r += 4) {
                const int h = h0 + r;
                if (h >= H) break;
                vec_t sg0, sg1;
                if constexpr (SJ) {                              // the join's gradient at this pixel, both batches: requested in front of the products
                    const e16* gp = sj.g + ib + ((long)h * T + (t < T ? t : T - 1)) * C;
                    sg0 = *reinterpret_cast<const vec_t*>(gp);
                    sg1 = *reinterpret_cast<const vec_t*>(gp + sj.half);
                }
                f32x4 a4[NB];
#pragma unroll
                for (int ob = 0; ob < NB; ++ob) a4[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int kh = tap / 3, kw = tap - 3 * kh;
                    const int pxi = (r + kh * D) * G::RW + G::DP + lane + (kw - 1) * D;
                    const vec_t bq = *reinterpret_cast<const vec_t*>(hs + (long)pxi * G::PXB);
#pragma unroll
                    for (int ob = 0; ob < NB; ++ob)
#pragma unroll
                        for (int kb = 0; kb < NB; ++kb) a4[ob] = mma4(A[tap][ob][kb], chunk_of<C>(bq, kb), a4[ob]);
                }
                const vec_t rq = *reinterpret_cast<const vec_t*>(gs + (long)((r + D) * G::RW + G::DP + lane) * G::PXB);
                vec_t o;
                if constexpr (GOUT) {
                    const vec_t xq = *reinterpret_cast<const vec_t*>(xs + (long)r * XROWB + lane * G::PXB);
                    if constexpr (SJ) {
                        float add[C], dot = 0.f;
                        skipj_terms<vec_t, C>(sg0, sg1, sj.half != 0, xq, sjw, add, dot);
                        if (t < T) sjdot += dot;
#pragma unroll
                        for (int c = 0; c < C; ++c) o[c] = (e16)((a4[c >> 2][c & 3] + (float)rq[c] + add[c]) * elu_dout((float)xq[c]));
                    } else {
#pragma unroll
                        for (int c = 0; c < C; ++c) o[c] = (e16)((a4[c >> 2][c & 3] + (float)rq[c]) * elu_dout((float)xq[c]));
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < C; ++c) o[c] = (e16)(a4[c >> 2][c & 3] + (float)rq[c]);
                }
                if (t < T) *reinterpret_cast<vec_t*>(dx + ib + ((long)h * T + t) * C) = o;
            }
        }

        // ---- phase 2b: dW1[tap] = sum over the tile's own pixels q of x[q] (x) dA1[q - off(tap)]: x from its halo-free image, dA1 at
        //      row r + (2 - kh) D, column DP - (kw - 1) D of the halo'd one; 32-byte slots by transpose reads as in k_nrb_wgrad (the two
        //      K halves are the two halves of a row at C = 8, two consecutive rows at C = 4) ----
        {
            constexpr int ROWB = G::RW * G::PXB;                 // bytes of a row of the halo'd images
            constexpr int RPS = C == 8 ? 1 : 2;                  // rows per product step
            constexpr int UOFF = C == 8 ? 512 : ROWB, UOFFX = C == 8 ? 512 : XROWB;    // byte distance of the second K half
            const int so = 32 * (4 * g + trj) + 8 * trq;
            for (int r = wave * RPS; r < G::TH; r += 4 * RPS) {
                if (h0 + r >= H) break;
                const unsigned char* xp = xs + (long)r * XROWB + so;
                const s16x4 xlo = lds_tr16(xp), xhi = lds_tr16(xp + UOFFX);
                const e16x8 xb8 = __builtin_bit_cast(e16x8, __builtin_shufflevector(xlo, xhi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const int kh = k / 3, kw = k - 3 * kh;
                    const unsigned char* gp = hs + (long)(r + (2 - kh) * D) * ROWB + (G::DP - (kw - 1) * D) * G::PXB + so;
                    const s16x4 lo = lds_tr16(gp), hi = lds_tr16(gp + UOFF);
                    wacc[k] = mma32(__builtin_bit_cast(e16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7)), xb8, wacc[k]);
                }
            }
        }
    }
    if constexpr (SJ) skipj_finish(sjdot, sj);
    // ---- dumps: the four waves hold the same weight-gradient elements -- summed through LDS, ONE dump per workgroup (wave slot 0;
    //      RedArgs::one_dump: the reduce reads a quarter of the bytes) ----
    __syncthreads();
    {
        float* wr = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) wr[wave * 2304 + (k * 4 + r) * 64 + lane] = wacc[k][r];
        __syncthreads();
        float* pw = part_w + (long)blockIdx.x * 4 * 2304;
        for (int i = tid; i < 2304; i += NT) pw[i] = (wr[i] + wr[2304 + i]) + (wr[2 * 2304 + i] + wr[3 * 2304 + i]);
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int e = 0; e < 2 * C; ++e) {
        float sv = acc[e];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sv += __shfl_xor(sv, o, 64);
        if (lane == 0) red[wave * ADUMP + C * C + e] = sv;
    }
    {
        float* sc = red + 4 * ADUMP + wave * 256;
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[(4 * g + r) * 16 + n] = dw2m[r];
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        if (lane < C * C) {
            const int co = lane / C, ci = lane - co * C;
            float sv = 0.f;
#pragma unroll
            for (int p_ = 0; p_ < 16 / C; ++p_) sv += sc[(p_ * C + co) * 16 + p_ * C + ci];
            red[wave * ADUMP + lane] = sv;
        }
    }
    __syncthreads();
    float* pa = part_a + (long)blockIdx.x * ADUMP;
    for (int e = tid; e < ADUMP; e += NT) pa[e] = (red[e] + red[ADUMP + e]) + (red[2 * ADUMP + e] + red[3 * ADUMP + e]);
}

// ---- data gradient and weight gradient of a narrow block in one pass (the narrow counterpart of k_wrb_dxw) --------------------
// Phases 2a / 2b of k_nrb_bwd_fused with dA1 coming from HBM (written by k_nrb_bwd_a) instead of being computed in place: the dA1
// and x tiles (16 x 64 pixels + halo) are staged once, dx = dy + W1^T (*) dA1 by lane-per-pixel 4x4x4 products, dW1 by transpose reads
// of 32-byte slots (k_nrb_wgrad's scheme; the two K halves are the two halves of a row at C = 8, two consecutive rows at C = 4).
template <int C, int D>
__global__ __launch_bounds__(NT) void k_nrb_dxw(const e16* __restrict__ x, const e16* __restrict__ da1, const e16* __restrict__ dy,
                                                const float* __restrict__ w1, e16* __restrict__ dx, float* __restrict__ part_w, int B,
                                                int H, int T, int tiles_h, int tiles_t, int ntiles) {
    using G = NTl<C, D>;
    typedef typename VecOf<C>::type vec_t;
    constexpr int NB = C / 4, IMG = G::NPR * 16;
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* hs = smem;                                    // dA1
    unsigned char* xs = smem + IMG;                              // x
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i4 = lane & 3;
    const int n = lane & 15, g = lane >> 4, trj = n >> 2, trq = n & 3;
    s16x4 A[9][NB][NB];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ob = 0; ob < NB; ++ob)
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {
                e16x4 t4;
#pragma unroll
                for (int k = 0; k < 4; ++k) t4[k] = (e16)w1[((4 * kb + k) * C + 4 * ob + i4) * 9 + (8 - tap)];
                A[tap][ob][kb] = __builtin_bit_cast(s16x4, t4);
            }
    f32x4 wacc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wacc[k] = f32x4{0.f, 0.f, 0.f, 0.f};

    const e16* zero = reinterpret_cast<const e16*>(&g_wzero16);
    for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
        int tile = xcd_order(v, ntiles);
        const int tt = tile % tiles_t; tile /= tiles_t;
        const int th = tile % tiles_h;
        const int b = tile / tiles_h, h0 = th * G::TH, t0 = tt * G::TW;
        const long ib = (long)b * H * T * C;
        __syncthreads();
        for (int i = wave * 64; i < G::NPR; i += NT) {
            const int p = i + lane, q = p * G::PPP;
            const int row = q / G::RW, px = q - row * G::RW;
            const int h = h0 - D + row, t = t0 - G::DP + px;
            const bool ok = p < G::NP && (unsigned)h < (unsigned)H && (unsigned)t < (unsigned)T;
            const long off = ib + ((long)h * T + t) * C;
            glds16(ok ? da1 + off : zero, hs + (long)i * 16);
            glds16(ok ? x + off : zero, xs + (long)i * 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        // ---- dx = dy + W1^T (*) dA1 ----
        {
            const int t = t0 + lane;
            const bool valid = t < T;
            for (int r = wave; r < G::TH; r += 4) {
                const int h = h0 + r;
                if (h >= H) break;
                const long pix = ((long)b * H + h) * T + t;
                const vec_t rq = *reinterpret_cast<const vec_t*>(dy + (valid ? pix : pix - (t - (T - 1))) * C);
                f32x4 a4[NB];
#pragma unroll
                for (int ob = 0; ob < NB; ++ob) a4[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int kh = tap / 3, kw = tap - 3 * kh;
                    const int pxi = (r + kh * D) * G::RW + G::DP + lane + (kw - 1) * D;
                    const vec_t bq = *reinterpret_cast<const vec_t*>(hs + (long)pxi * G::PXB);
#pragma unroll
                    for (int ob = 0; ob < NB; ++ob)
#pragma unroll
                        for (int kb = 0; kb < NB; ++kb) a4[ob] = mma4(A[tap][ob][kb], chunk_of<C>(bq, kb), a4[ob]);
                }
                vec_t o;
#pragma unroll
                for (int c = 0; c < C; ++c) o[c] = (e16)(a4[c >> 2][c & 3] + (float)rq[c]);
                if (valid) *reinterpret_cast<vec_t*>(dx + pix * C) = o;
            }
        }
        // ---- dW1 ----
        {
            constexpr int ROWB = G::RW * G::PXB;                 // bytes of an image row
            constexpr int RPS = C == 8 ? 1 : 2;                  // rows per product step
            constexpr int UOFF = C == 8 ? 512 : ROWB;            // byte distance of the second K half
            const int so = 32 * (4 * g + trj) + 8 * trq;
            for (int r = wave * RPS; r < G::TH; r += 4 * RPS) {
                if (h0 + r >= H) break;
                const unsigned char* gp = hs + (long)(r + D) * ROWB + G::DP * G::PXB + so;
                const s16x4 glo = lds_tr16(gp), ghi = lds_tr16(gp + UOFF);
                const e16x8 ga = __builtin_bit_cast(e16x8, __builtin_shufflevector(glo, ghi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const int kh = k / 3, kw = k - 3 * kh;
                    const unsigned char* xp = xs + (long)(r + kh * D) * ROWB + (G::DP + (kw - 1) * D) * G::PXB + so;
                    const s16x4 lo = lds_tr16(xp), hi = lds_tr16(xp + UOFF);
                    wacc[k] = mma32(ga, __builtin_bit_cast(e16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7)), wacc[k]);
                }
            }
        }
    }
    // the four waves hold the same elements: summed through LDS, ONE dump per workgroup (RedArgs::one_dump)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave * 2304 + (k * 4 + r) * 64 + lane] = wacc[k][r];
    __syncthreads();
    float* pw = part_w + (long)blockIdx.x * 4 * 2304;
    for (int i = tid; i < 2304; i += NT) pw[i] = (red[i] + red[2304 + i]) + (red[2 * 2304 + i] + red[3 * 2304 + i]);
}

template <int C>
__global__ __launch_bounds__(1024) void k_nrb_reduce(RedBatch batch) {
    const RedArgs& ar = batch.a[blockIdx.y];
    constexpr int WDUMP = 9 * 256, ADUMP = C * C + 2 * C;
    __shared__ float red[RSL][REL];
    const int el = threadIdx.x % REL, sl = threadIdx.x / REL;
    const int e = blockIdx.x * REL + el;
    float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (e < WDUMP) {
        const int stride = ar.one_dump ? 4 : 1;                  // one_dump: only wave slot 0 of every workgroup holds data
        const int ncontrib = ar.one_dump ? ar.gw : ar.gw * 4;
        for (int j0 = sl; j0 < ncontrib; j0 += 8 * RSL)
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (j0 + RSL * u < ncontrib) part[u] += ar.pw[(long)(j0 + RSL * u) * stride * WDUMP + e];
    } else if (e < WDUMP + ADUMP) {
        const int ncontrib = ar.ga;
        for (int j0 = sl; j0 < ncontrib; j0 += 8 * RSL)
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (j0 + RSL * u < ncontrib) part[u] += ar.pa[(long)(j0 + RSL * u) * ADUMP + (e - WDUMP)];
    }
    red[sl][el] = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
    __syncthreads();
    if (sl != 0) return;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < RSL; ++i) sum += red[i][el];
    if (e < WDUMP) {
        const int k = e >> 8, r = (e >> 6) & 3, lane = e & 63, g = lane >> 4, n = lane & 15;
        // D row 4g + r = (pixel-in-slot s, channel c), column n = (s', c'): keep s == s'
        const int m = 4 * g + r, s = m / C, c = m - s * C, s2 = n / C, c2 = n - s2 * C;
        if (s == s2) atomicAdd(ar.dw1 + (c * C + c2) * 9 + k, sum * ar.scale);   // 16 / C contributions per element
    } else if (e < WDUMP + ADUMP) {
        const int q = e - WDUMP;
        float* dst = q < C * C ? ar.dw2 + q : (q < C * C + C ? ar.db1 + (q - C * C) : ar.db2 + (q - C * C - C));
        *dst += sum * ar.scale;
    }
}

template <int C, int D, int MODE, bool SAVE>
int launch_nconv(const e16* x, const float* w1, const float* b1, const float* w2, const float* b2, const e16* res,
                 e16* y, e16* h1, int B, int H, int T, hipStream_t st, const float* jw = nullptr, int jB = 1) {
    using G = NTl<C, D>;
    static AttrOnce once;
    auto kern = k_nrb_conv<C, D, MODE, SAVE>;
    if (int rc = raise_lds(kern, G::LDS_BYTES, once)) return rc;
    const int tiles_h = (H + G::TH - 1) / G::TH, tiles_t = (T + G::TW - 1) / G::TW, ntiles = B * tiles_h * tiles_t;
    // workgroups per CU (LDS allows 5-10): 2 / 4 / 6 / 8 -> data gradient of a C = 4 block 0.25 / 0.19 / 0.165 / 0.165 ms
    static const int per_cu = tt_tune("TTRAP_NARROW_PER_CU", 6);
    hipLaunchKernelGGL(kern, dim3(grid_for(ntiles, G::LDS_BYTES, per_cu)), dim3(NT), G::LDS_BYTES, st, x, w1, b1, w2, b2, res, y, h1,
                       B, H, T, tiles_h, tiles_t, ntiles, jw, jB);
    TT_LAUNCH_CHECK();
    return 0;
}

template <int C> constexpr long ndump_floats() { return (long)MAX_A_WG * (C * C + 2 * C) + (long)MAX_W_WG * 4 * 9 * 256; }

template <int C, int D>
int launch_nbwd(const e16* x, const e16* h1, const e16* dy, const float* w1, const float* w2, const float* b2,
                e16* dx, float* dw1, float* db1, float* dw2, float* db2, unsigned char* ws, int B, int H, int T,
                hipStream_t st) {
    const long npix = (long)B * H * T;
    e16* da1 = reinterpret_cast<e16*>(ws);
    float* part_a = reinterpret_cast<float*>(ws + ((npix * C * 2 + 255) / 256) * 256);
    float* part_w = part_a + (long)MAX_A_WG * (C * C + 2 * C);
    // Measured per launch at the bench shapes, pointwise + merged gradient kernels -> fused (with the weight image in LDS):
    // C = 8: 0.410 / 0.409 / 0.452 -> 0.394 / 0.411 / 0.563 ms for dilation 1 / 2 / 3, C = 4: 0.40 / 0.407 / 0.408 -> 0.338 / 0.344 / 0.383 ms.
    // Half the HBM traffic buys less than half the time because the pass is then bound by its own arithmetic (pointwise chain on the halo
    // as well; three LDS images = 2 workgroups per CU at C = 8): fused where it wins -- C = 4 always, C = 8 at dilation 1, 2.
    static const int fused = tt_switch("TTRAP_NARROW_FUSED16", 1);
    if (fused) {
        using F = NTl<C, D>;
        constexpr int LAYOUT = 2 * ((F::NP * 16 + 63) / 64 * 64) + F::TH * F::TW * F::PXB + 11 * (C / 4) * (C / 4) * 32 + 4 * 64 * F::PXB;
        constexpr int LDS = LAYOUT > 4 * 2304 * 4 ? LAYOUT : 4 * 2304 * 4;       // the epilogue sums the four waves' dW1 accumulators through LDS
        const int tiles_h = (H + F::TH - 1) / F::TH, tiles_t = (T + F::TW - 1) / F::TW, ntiles = B * tiles_h * tiles_t;
        int gf = grid_for(ntiles, LDS, 4);
        if (gf > MAX_W_WG) gf = MAX_W_WG;
        bool gated = false;
        if constexpr (D == 1) gated = ttx_gate_dx == 1;          // a level's first block (the reference's levels start at dilation 1)
        if (gated) {
            if constexpr (D == 1) {
                if (ttx_skip.g) {
                    static AttrOnce once_j;
                    auto kj = k_nrb_bwd_fused<C, D, true, true>;
                    if (int rc = raise_lds(kj, LDS, once_j)) return rc;
                    hipLaunchKernelGGL(kj, dim3(gf), dim3(NT), LDS, st, x, h1, dy, w1, w2, b2, dx, part_a, part_w, B, H, T, tiles_h, tiles_t, ntiles, ttx_skip);
                    ttx_skip.g = nullptr;
                } else {
                    static AttrOnce once_g;
                    auto kg = k_nrb_bwd_fused<C, D, true>;
                    if (int rc = raise_lds(kg, LDS, once_g)) return rc;
                    hipLaunchKernelGGL(kg, dim3(gf), dim3(NT), LDS, st, x, h1, dy, w1, w2, b2, dx, part_a, part_w, B, H, T, tiles_h, tiles_t, ntiles, SkipJ());
                }
                ttx_gate_dx = 2;
            }
        } else {
            static AttrOnce once_f;
            auto kf = k_nrb_bwd_fused<C, D, false>;
            if (int rc = raise_lds(kf, LDS, once_f)) return rc;
            hipLaunchKernelGGL(kf, dim3(gf), dim3(NT), LDS, st, x, h1, dy, w1, w2, b2, dx, part_a, part_w, B, H, T, tiles_h, tiles_t, ntiles, SkipJ());
        }
        TT_LAUNCH_CHECK();
        RedArgs ra{part_w, gf, part_a, gf, dw1, db1, dw2, db2, 0, 1};
        constexpr int total = 9 * 256 + C * C + 2 * C;
        return reduce_or_defer(k_nrb_reduce<C>, total, ra, st);
    }
    const long ngroups = (npix + 63) / 64;
    const long want = (ngroups + 3) / 4;
    static const int a_per_cu = tt_tune("TTRAP_BWDA_PER_CU", 4);
    int grid = (int)(want < (long)a_per_cu * tt_cus() ? want : (long)a_per_cu * tt_cus());
    if (grid > MAX_A_WG) grid = MAX_A_WG;
    hipLaunchKernelGGL(k_nrb_bwd_a<C>, dim3(grid), dim3(NT), 0, st, h1, dy, w2, b2, da1, part_a, npix, ngroups);
    TT_LAUNCH_CHECK();
    static const int ndxw = tt_switch("TTRAP_NDXW", 1);
    if (ndxw) {                                                  // data + weight gradient in one pass
        using F = NTl<C, D>;
        constexpr int LDS = 2 * F::NPR * 16 > 4 * 2304 * 4 ? 2 * F::NPR * 16 : 4 * 2304 * 4;   // images, then the four waves' accumulators
        static AttrOnce once_x;
        auto kx = k_nrb_dxw<C, D>;
        if (int rc = raise_lds(kx, LDS, once_x)) return rc;
        const int tiles_h = (H + F::TH - 1) / F::TH, tiles_t = (T + F::TW - 1) / F::TW, ntiles = B * tiles_h * tiles_t;
        static const int x_per_cu = tt_tune("TTRAP_NDXW_PER_CU", 4);
        int gx = grid_for(ntiles, LDS, x_per_cu);
        if (gx > MAX_W_WG) gx = MAX_W_WG;
        hipLaunchKernelGGL(kx, dim3(gx), dim3(NT), LDS, st, x, da1, dy, w1, dx, part_w, B, H, T, tiles_h, tiles_t, ntiles);
        TT_LAUNCH_CHECK();
        RedArgs ra{part_w, gx, part_a, grid, dw1, db1, dw2, db2, 0, 1};
        constexpr int total = 9 * 256 + C * C + 2 * C;
        return reduce_or_defer(k_nrb_reduce<C>, total, ra, st);
    }
    if (int rc = launch_nconv<C, D, 1, false>(da1, w1, nullptr, nullptr, nullptr, dy, dx, nullptr, B, H, T, st)) return rc;
    using W = NG<C, D>;
    static AttrOnce once_w;
    auto kw = k_nrb_wgrad<C, D>;
    if (int rc = raise_lds(kw, W::LDS_BYTES, once_w)) return rc;
    const int tiles_h = (H + W::TH - 1) / W::TH, tiles_t = (T + W::TW - 1) / W::TW, ntiles = B * tiles_h * tiles_t;
    static const int w_per_cu = tt_tune("TTRAP_NWGRAD_PER_CU", 3);
    int gw = grid_for(ntiles, W::LDS_BYTES, w_per_cu);
    if (gw > MAX_W_WG) gw = MAX_W_WG;
    hipLaunchKernelGGL(kw, dim3(gw), dim3(NT), W::LDS_BYTES, st, x, da1, part_w, B, H, T, tiles_h, tiles_t, ntiles);
    TT_LAUNCH_CHECK();
    RedArgs ra{part_w, gw, part_a, grid, dw1, db1, dw2, db2};
    constexpr int total = 9 * 256 + C * C + 2 * C;
    return reduce_or_defer(k_nrb_reduce<C>, total, ra, st);
}

template <int C>
int nfwd_c(const e16* x, const float* w1, const float* b1, const float* w2, const float* b2, e16* y, e16* h1, int B,
           int H, int T, int D, hipStream_t st, const e16* je = nullptr, const float* jw = nullptr, int jB = 1) {
#define TT_NFWD(DD)                                                                                                  \
    if (je) return h1 ? launch_nconv<C, DD, 2, true>(x, w1, b1, w2, b2, je, y, h1, B, H, T, st, jw, jB)               \
                      : launch_nconv<C, DD, 2, false>(x, w1, b1, w2, b2, je, y, nullptr, B, H, T, st, jw, jB);        \
    return h1 ? launch_nconv<C, DD, 0, true>(x, w1, b1, w2, b2, nullptr, y, h1, B, H, T, st)                          \
              : launch_nconv<C, DD, 0, false>(x, w1, b1, w2, b2, nullptr, y, nullptr, B, H, T, st)
    switch (D) {
        case 1: TT_NFWD(1);
        case 2: TT_NFWD(2);
        case 3: TT_NFWD(3);
    }
#undef TT_NFWD
    return TT_E_UNSUPPORTED;
}
template <int C>
int nbwd_c(const e16* x, const e16* h1, const e16* dy, const float* w1, const float* w2, const float* b2, e16* dx,
           float* dw1, float* db1, float* dw2, float* db2, unsigned char* ws, int B, int H, int T, int D, hipStream_t st) {
    switch (D) {
        case 1: return launch_nbwd<C, 1>(x, h1, dy, w1, w2, b2, dx, dw1, db1, dw2, db2, ws, B, H, T, st);
        case 2: return launch_nbwd<C, 2>(x, h1, dy, w1, w2, b2, dx, dw1, db1, dw2, db2, ws, B, H, T, st);
        case 3: return launch_nbwd<C, 3>(x, h1, dy, w1, w2, b2, dx, dw1, db1, dw2, db2, ws, B, H, T, st);
    }
    return TT_E_UNSUPPORTED;
}

inline bool shape_ok(int B, int C, int H, int T) {
    return B > 0 && H > 0 && T > 0 && (C == 4 || C == 8 || C == 16 || C == 32) && (C != 4 || T % 2 == 0) &&
           (long)H * T * C < (1l << 31);
}

// dx *= ELU'(x) in place, 8 elements per thread step: what tt_wide_level_bwd_gated runs behind a first block whose backward kernel has no
// gated form (a dilation other than 1, the A/B paths behind TTRAP_DXW / TTRAP_NARROW_FUSED16 / TTRAP_WBWD1)
__global__ __launch_bounds__(256) void k_gate_dx(e16* __restrict__ dx, const e16* __restrict__ x, long n8) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        e16x8 d = reinterpret_cast<const e16x8*>(dx)[i];
        const e16x8 v = reinterpret_cast<const e16x8*>(x)[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = (e16)((float)d[j] * elu_dout((float)v[j]));
        reinterpret_cast<e16x8*>(dx)[i] = d;
    }
}

}  // namespace

thread_local void* ttx_red_defer = nullptr;
thread_local bool ttx_wprep_done = false;
thread_local SkipJ ttx_skip;
thread_local int ttx_gate_dx = 0;             // 1: the next block backward should leave dx * ELU'(x); 2: its kernel did (wide_common.h)

extern "C" {

int64_t tt_wide_scratch_bytes(int B, int C, int H, int T) {
    if (!shape_ok(B, C, H, T)) return -1;
    const long npix = (long)B * H * T;
    const long dumps = C == 4 ? ndump_floats<4>() : C == 8 ? ndump_floats<8>() : C == 16 ? dump_floats<16>() : dump_floats<32>();
    return ((npix * C * 2 + 255) / 256) * 256 + dumps * 4;
}

int tt_wide_pack(const float* x, void* out, int B, int C, int H, int T, void* stream) {
    if (!x || !out || !(shape_ok(B, C, H, T) || (C == 64 && B > 0 && H > 0 && T > 0))) return TT_E_BADARG;
    const long npix = (long)B * H * T, pieces = npix * C / 8;
    const unsigned grid = (unsigned)((pieces + NT - 1) / NT);
    hipStream_t st = tt_stream(stream);
    switch (C) {
        case 4: hipLaunchKernelGGL(k_wide_pack<4>, dim3(grid), dim3(NT), 0, st, x, (e16*)out, H, T, npix); break;
        case 8: hipLaunchKernelGGL(k_wide_pack<8>, dim3(grid), dim3(NT), 0, st, x, (e16*)out, H, T, npix); break;
        case 16: hipLaunchKernelGGL(k_wide_pack<16>, dim3(grid), dim3(NT), 0, st, x, (e16*)out, H, T, npix); break;
        case 64: hipLaunchKernelGGL(k_wide_pack<64>, dim3(grid), dim3(NT), 0, st, x, (e16*)out, H, T, npix); break;
        default: hipLaunchKernelGGL(k_wide_pack<32>, dim3(grid), dim3(NT), 0, st, x, (e16*)out, H, T, npix);
    }
    TT_LAUNCH_CHECK();
    return 0;
}

int tt_wide_unpack(const void* in, float* y, int B, int C, int H, int T, void* stream) {
    if (!in || !y || !(shape_ok(B, C, H, T) || (C == 64 && B > 0 && H > 0 && T > 0))) return TT_E_BADARG;
    const long npix = (long)B * H * T, pieces = npix * C / 8;
    const unsigned grid = (unsigned)((pieces + NT - 1) / NT);
    hipStream_t st = tt_stream(stream);
    switch (C) {
        case 4: hipLaunchKernelGGL(k_wide_unpack<4>, dim3(grid), dim3(NT), 0, st, (const e16*)in, y, H, T, npix); break;
        case 8: hipLaunchKernelGGL(k_wide_unpack<8>, dim3(grid), dim3(NT), 0, st, (const e16*)in, y, H, T, npix); break;
        case 16: hipLaunchKernelGGL(k_wide_unpack<16>, dim3(grid), dim3(NT), 0, st, (const e16*)in, y, H, T, npix); break;
        case 64: hipLaunchKernelGGL(k_wide_unpack<64>, dim3(grid), dim3(NT), 0, st, (const e16*)in, y, H, T, npix); break;
        default: hipLaunchKernelGGL(k_wide_unpack<32>, dim3(grid), dim3(NT), 0, st, (const e16*)in, y, H, T, npix);
    }
    TT_LAUNCH_CHECK();
    return 0;
}

int tt_wide_rb_fwd(const void* x, const float* w1, const float* b1, const float* w2, const float* b2, void* y, void* h1, int B,
                   int C, int H, int T, int dilation, void* stream) {
    if (!x || !w1 || !b1 || !w2 || !b2 || !y || !shape_ok(B, C, H, T)) return TT_E_BADARG;
    const e16* xi = (const e16*)x;
    e16 *yo = (e16*)y, *ho = (e16*)h1;
    hipStream_t st = tt_stream(stream);
    switch (C) {
        case 4: return nfwd_c<4>(xi, w1, b1, w2, b2, yo, ho, B, H, T, dilation, st);
        case 8: return nfwd_c<8>(xi, w1, b1, w2, b2, yo, ho, B, H, T, dilation, st);
        case 16: return fwd_c<16>(xi, w1, b1, w2, b2, yo, ho, B, H, T, dilation, st);
    }
    return fwd_c<32>(xi, w1, b1, w2, b2, yo, ho, B, H, T, dilation, st);
}

int tt_wide_rb_fwd_join(const void* x, const float* w1, const float* b1, const float* w2, const float* b2, void* y, void* h1,
                        const void* skip, const float* skip_weights, int skip_idx, int skip_B, int B, int C, int H, int T, int dilation,
                        void* stream) {
    if (!x || !w1 || !b1 || !w2 || !b2 || !y || !skip || skip_idx < 0 || skip_B < 1 || B % skip_B || !shape_ok(B, C, H, T)) return TT_E_BADARG;
    if (y == skip || y == x) return TT_E_BADARG;
    const e16 *xi = (const e16*)x, *je = (const e16*)skip;
    const float* jw = skip_weights ? skip_weights + skip_idx : nullptr;
    e16 *yo = (e16*)y, *ho = (e16*)h1;
    hipStream_t st = tt_stream(stream);
    switch (C) {
        case 4: return nfwd_c<4>(xi, w1, b1, w2, b2, yo, ho, B, H, T, dilation, st, je, jw, skip_B);
        case 8: return nfwd_c<8>(xi, w1, b1, w2, b2, yo, ho, B, H, T, dilation, st, je, jw, skip_B);
        case 16: return fwd_c<16>(xi, w1, b1, w2, b2, yo, ho, B, H, T, dilation, st, je, jw, skip_B);
    }
    return fwd_c<32>(xi, w1, b1, w2, b2, yo, ho, B, H, T, dilation, st, je, jw, skip_B);
}

int tt_wide_rb_bwd_is_onepass(int C, int dilation) {
    return (C == 16 || C == 32) && dilation >= 1 && dilation <= 3 ? onepass_choice(C, dilation) : 0;
}

int tt_wide_rb_bwd(const void* x, const void* h1, const void* dy, const float* w1, const float* w2, const float* b2, void* dx,
                   float* dw1, float* db1, float* dw2, float* db2, void* ws, int B, int C, int H, int T, int dilation,
                   void* stream) {
    if (!x || !h1 || !dy || !w1 || !w2 || !b2 || !dx || !dw1 || !db1 || !dw2 || !db2 || !ws || !shape_ok(B, C, H, T))
        return TT_E_BADARG;
    const e16 *xi = (const e16*)x, *hi = (const e16*)h1, *gi = (const e16*)dy;
    e16* go = (e16*)dx;
    unsigned char* w = (unsigned char*)ws;
    hipStream_t st = tt_stream(stream);
    switch (C) {
        case 4: return nbwd_c<4>(xi, hi, gi, w1, w2, b2, go, dw1, db1, dw2, db2, w, B, H, T, dilation, st);
        case 8: return nbwd_c<8>(xi, hi, gi, w1, w2, b2, go, dw1, db1, dw2, db2, w, B, H, T, dilation, st);
        case 16: return bwd_c<16>(xi, hi, gi, w1, w2, b2, go, dw1, db1, dw2, db2, w, B, H, T, dilation, st);
    }
    return bwd_c<32>(xi, hi, gi, w1, w2, b2, go, dw1, db1, dw2, db2, w, B, H, T, dilation, st);
}

int64_t tt_wide_level_scratch_bytes(int nblocks, int B, int C, int H, int T) {
    const int64_t one = tt_wide_scratch_bytes(B, C, H, T);
    return one < 0 || nblocks < 1 || nblocks > 4 ? -1 : (int64_t)nblocks * ((one + 255) / 256 * 256);
}

static int level_bwd(int gate, int nblocks, const void* const* x, const void* const* h1, const void* dy, const float* const* w1,
                     const float* const* w2, const float* const* b2, void* dx, void* tmp0, void* tmp1, float* const* dw1,
                     float* const* db1, float* const* dw2, float* const* db2, void* ws, int B, int C, int H, int T,
                     const int* dilations, void* stream) {
    if (nblocks < 1 || nblocks > 4 || !x || !h1 || !dy || !w1 || !w2 || !b2 || !dx || !dw1 || !db1 || !dw2 || !db2 || !ws || !dilations ||
        (nblocks > 1 && (!tmp0 || !tmp1)) || !shape_ok(B, C, H, T))
        return TT_E_BADARG;
    if (ttx_red_defer) return TT_E_BADARG;                      // not re-entrant on one thread
    const int64_t one = (tt_wide_scratch_bytes(B, C, H, T) + 255) / 256 * 256;
    RedBatch batch;
    int rc = 0;
    // the weight images of the blocks that take the one-pass kernel: one launch for the level
    {
        const float *pw1[4], *pw2[4];
        void* pws[4];
        int np = 0;
        for (int i = 0; i < nblocks; ++i)
            if (tt_wide_rb_bwd_is_onepass(C, dilations[i])) { pw1[np] = w1[i]; pw2[np] = w2[i]; pws[np] = (unsigned char*)ws + (int64_t)i * one; ++np; }
        if (np > 0) {
            if ((rc = ttx_wide_wprep_batch(C, np, pw1, pw2, pws, tt_stream(stream)))) return rc;
            ttx_wprep_done = true;
        }
    }
    ttx_red_defer = &batch;
    const void* g = dy;
    for (int i = nblocks - 1; i >= 0 && !rc; --i) {
        void* gx = i == 0 ? dx : ((i & 1) ? tmp1 : tmp0);
        ttx_gate_dx = (i == 0 && gate) ? 1 : 0;
        rc = tt_wide_rb_bwd(x[i], h1[i], g, w1[i], w2[i], b2[i], gx, dw1[i], db1[i], dw2[i], db2[i], (unsigned char*)ws + (int64_t)i * one, B, C, H, T,
                            dilations[i], stream);
        g = gx;
    }
    const bool gate_left = ttx_gate_dx == 1;                    // asked for and not done by the first block's kernel
    ttx_gate_dx = 0;
    ttx_red_defer = nullptr;
    ttx_wprep_done = false;
    if (rc) return rc;
    hipStream_t st = tt_stream(stream);
    if (gate_left) {
        const long n8 = (long)B * H * T * C / 8;                 // C * T is a multiple of 8 (shape_ok)
        const long want = (n8 + 255) / 256, cap = (long)8 * tt_cus();
        hipLaunchKernelGGL(k_gate_dx, dim3((unsigned)(want < cap ? want : cap)), dim3(256), 0, st, (e16*)dx, (const e16*)x[0], n8);
        TT_LAUNCH_CHECK();
    }
    if (batch.n > 0) {
        const int total = (C <= 8 ? 9 * 256 : 9 * C * C) + C * C + 2 * C;
        const dim3 grid((total + REL - 1) / REL, batch.n);
        switch (C) {
            case 4: hipLaunchKernelGGL(k_nrb_reduce<4>, grid, dim3(1024), 0, st, batch); break;
            case 8: hipLaunchKernelGGL(k_nrb_reduce<8>, grid, dim3(1024), 0, st, batch); break;
            case 16: hipLaunchKernelGGL(k_wrb_reduce<16>, grid, dim3(1024), 0, st, batch); break;
            default: hipLaunchKernelGGL(k_wrb_reduce<32>, grid, dim3(1024), 0, st, batch);
        }
        TT_LAUNCH_CHECK();
    }
    return 0;
}

int tt_gate16(void* g, const void* y, int64_t n, void* stream) {
    if (!g || !y || n < 0 || n % 8) return TT_E_BADARG;
    if (n == 0) return 0;
    const long n8 = n / 8, want = (n8 + 255) / 256, cap = (long)8 * tt_cus();
    hipLaunchKernelGGL(k_gate_dx, dim3((unsigned)(want < cap ? want : cap)), dim3(256), 0, tt_stream(stream), (e16*)g, (const e16*)y, n8);
    TT_LAUNCH_CHECK();
    return 0;
}

int tt_wide_level_bwd(int nblocks, const void* const* x, const void* const* h1, const void* dy, const float* const* w1,
                      const float* const* w2, const float* const* b2, void* dx, void* tmp0, void* tmp1, float* const* dw1,
                      float* const* db1, float* const* dw2, float* const* db2, void* ws, int B, int C, int H, int T,
                      const int* dilations, void* stream) {
    return level_bwd(0, nblocks, x, h1, dy, w1, w2, b2, dx, tmp0, tmp1, dw1, db1, dw2, db2, ws, B, C, H, T, dilations, stream);
}

int tt_wide_level_bwd_gated(int nblocks, const void* const* x, const void* const* h1, const void* dy, const float* const* w1,
                            const float* const* w2, const float* const* b2, void* dx, void* tmp0, void* tmp1, float* const* dw1,
                            float* const* db1, float* const* dw2, float* const* db2, void* ws, int B, int C, int H, int T,
                            const int* dilations, void* stream) {
    return level_bwd(1, nblocks, x, h1, dy, w1, w2, b2, dx, tmp0, tmp1, dw1, db1, dw2, db2, ws, B, C, H, T, dilations, stream);
}

int tt_wide_level_bwd_gated_join(int nblocks, const void* const* x, const void* const* h1, const void* dy, const float* const* w1,
                                 const float* const* w2, const float* const* b2, void* dx, void* tmp0, void* tmp1, float* const* dw1,
                                 float* const* db1, float* const* dw2, float* const* db2, void* ws, int B, int C, int H, int T,
                                 const int* dilations, const void* skip_g, int skip_reps, const float* skip_weights, int skip_idx,
                                 float* skip_dw, void* stream) {
    if (!skip_g || (skip_reps != 1 && skip_reps != 2) || skip_idx < 0 || !dilations || nblocks < 1) return TT_E_BADARG;
    if (dilations[0] != 1) return TT_E_UNSUPPORTED;              // only a first block at dilation 1 has a gated epilogue (the reference's levels)
    ttx_skip.g = (const e16*)skip_g;
    ttx_skip.half = skip_reps == 2 ? (long)B * H * T * C : 0;
    ttx_skip.w = skip_weights ? skip_weights + skip_idx : nullptr;
    ttx_skip.dw = skip_dw ? skip_dw + skip_idx : nullptr;
    ttx_skip.unscale = tt_loss_unscale();
    const int rc = level_bwd(1, nblocks, x, h1, dy, w1, w2, b2, dx, tmp0, tmp1, dw1, db1, dw2, db2, ws, B, C, H, T, dilations, stream);
    const bool left = ttx_skip.g != nullptr;                     // the first block took a kernel without the riding form (an A/B dispatch)
    ttx_skip = SkipJ();
    if (rc) return rc;
    return left ? TT_W_JOIN_LEFT : 0;
}

}  // extern "C"
