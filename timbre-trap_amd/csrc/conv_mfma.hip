// Implicit-GEMM convolutions of the Timbre-Trap autoencoder on the gfx950 fp32 matrix cores
// (v_mfma_f32_16x16x4_f32: exact fp32 -- bitwise an fmaf chain -- at the fp32 vector rate, fed by two LDS
// reads per 32-cycle instruction).
//
//   out[m][r][t] = sum_{tap} sum_{c} W[tap][c][m] * in[c][row(r, tap)][t + col(tap)]
//
// One main loop serves every layer geometry through a policy:
//   Res3x3<D>  ResidualConv2dBlock 3x3 conv, dilation D on H and T        (reference modules.py:746)
//   Down4      Conv2d (4,1) stride (2,1)  and the data gradient of Up4    (modules.py:628)
//   Up4        ConvTranspose2d (4,1) stride (2,1)  and the data gradient of Down4   (modules.py:687)
// MFMA mapping (16 x 16 x 4):
//   A fragment  lane l : Wimg[k0 + (l>>4)][mt*16 + (l&15)]     weights, transposed once into LDS, 16-column
//                                                                halves XOR-swizzled by k&1 (conflict-free at pitch 32/64)
//   B fragment  lane l : xs[c = .. + (l>>4)][row][col + (l&15)]  input tile + halo in LDS, plane pitch 16 / 17 mod 32
//   D fragment  lane l : rows 4*(l>>4) + r (r = 0..3), column l&15
// Workgroup = 8 waves = 8 output rows x 64 columns of one clip.  The input is staged per chunk of CC channels,
// double-buffered, ONE barrier per chunk: by LDS-DMA (global_load_lds_dwordx4, no registers) when rows are 16-byte
// aligned and T % 4 == 0, else through registers (any T; optional ELU' gating while staging).  Workgroups are
// persistent over tiles (weight image built once) and walk them in an XCD-contiguous order (xcd_tile).
// Epilogues address with a wave-uniform 64-bit base per (clip, channel) plus 32-bit lane offsets and move 16 bytes per
// lane (quad-transposed accumulators on the fp32 path).
//
// ResidualConv2dBlock (modules.py:755-777):  y = ELU(W2 . ELU(W1 (*) x + b1) + b2) + x
//   forward   : k_rb_fwd -- D fragments of the 3x3 stage hold, for register r, channels {4g + r} over the four lane
//               groups g: exactly a B fragment (k = g) for the 1x1 stage, the hidden activation never leaves registers
//               (it is optionally stored for the backward pass).
//   backward  : k_rb_bwd_a (pointwise chain from the saved or recomputed hidden activation -> dA1, db1, db2, dW2),
//               k_conv_mfma with flipped weights (dx = dy + W1^T (*) dA1), k_wgrad_dma / k_wgrad3_pack (dW1 as MFMA GEMM
//               with K = pixels; k_wgrad_mfma is the register-staged fallback).
// Precision of the wide 3x3 convs (PREC): 0 = fp32 (default), 1 = bf16 operands, 2 = split-bf16 (hi + lo, three
// v_mfma_f32_16x16x32_bf16 per product block); the bf16 modes use conv_mainloop_z (channel-interleaved bf16 tile).
// Narrow strided / transposed layers run on the vector ALUs from LDS-DMA staged tiles (k_conv_valu).
#include <cstdlib>
#include "common.h"
#include "conv_small.h"

#ifndef TT_RES_LDS
#define TT_RES_LDS 1      // 0: re-read the residual from memory in the epilogue (A/B builds)
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// bf16-operand mode (flags bit 0 of the resblock entry points): v_mfma_f32_16x16x32_bf16, fp32 accumulate.
//   A fragment lane l : 8 consecutive k (k = 8 (l>>4) + j) of row l&15 ;  B fragment: the same k of column l&15.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mfma16_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 pack_bf16(const float (&v)[8]) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (__bf16)v[j];
    return r;
}

// split-bf16 mode (flags bit 1, "bf16x3"): every fp32 operand is written as hi + lo with hi = bf16(v), lo = bf16(v - hi)
// (both round-to-nearest, |v - hi - lo| <= 2^-18 |v|) and a product is accumulated as hi*hi + lo*hi + hi*lo in fp32:
// fp32-class results (relative error of a product ~1e-5, unbiased) from three bf16 matrix instructions, 16/3 of the
// fp32 MFMA rate.
template <int PREC>
__device__ __forceinline__ void split_bf16(const float (&v)[8], bf16x8& h, bf16x8& l) {
    h = pack_bf16(v);
    if constexpr (PREC == 2) {
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = v[j] - (float)h[j];
        l = pack_bf16(r);
    }
}
template <int PREC>
__device__ __forceinline__ f32x4 mma_bf16(const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl, f32x4 c) {
    if constexpr (PREC == 2) {
        c = mfma16_bf16(al, bh, c);
        c = mfma16_bf16(ah, bl, c);
    }
    return mfma16_bf16(ah, bh, c);
}

// 4 x 4 transpose between the four lanes of a quad and four registers: afterwards v[i] of quad lane j holds what v[j] of
// quad lane i held.  Applied to the accumulators nt = 0..3 of one output channel (pixel nt * 16 + l15) it leaves lane
// l15 = 4 q + j with the four CONSECUTIVE pixels 16 j + 4 q + i: 16-byte epilogue accesses on the fp32 path as well.
__device__ __forceinline__ float dpp_quad_xor1(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float dpp_quad_xor2(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
}
__device__ __forceinline__ void quad_transpose(float (&v)[4], int lane) {
    const bool hi2 = lane & 2, hi1 = lane & 1;
    float a = hi2 ? v[0] : v[2], b = hi2 ? v[1] : v[3];
    a = dpp_quad_xor2(a); b = dpp_quad_xor2(b);
    if (hi2) { v[0] = a; v[1] = b; } else { v[2] = a; v[3] = b; }
    a = hi1 ? v[0] : v[1]; b = hi1 ? v[2] : v[3];
    a = dpp_quad_xor1(a); b = dpp_quad_xor1(b);
    if (hi1) { v[0] = a; v[2] = b; } else { v[1] = a; v[3] = b; }
}

constexpr int plane_pad(int n) {
    int p = n;
    while (p % 32 != 17) ++p;
    return p;
}
constexpr int cmax(int a, int b) { return a > b ? a : b; }

constexpr int NTHREADS = 512;   // 8 waves
constexpr int TH = 8, TW = 64;

// ---- geometry policies --------------------------------------------------------------------------------
template <int D>
struct Res3x3 {
    static constexpr int NTAPS = 9, NWT = 9, XR = TH + 2 * D, XC = TW + 2 * D, CH = D;
    static __device__ __forceinline__ int in_row0(int h0) { return h0 - D; }
    static __device__ __forceinline__ int lrow(int tp, int wave) { return wave + (tp / 3) * D; }
    static __device__ __forceinline__ int lcol(int tp) { return (tp % 3) * D; }
    static __device__ __forceinline__ int wtap(int tp, int) { return tp; }
};
struct Down4 {      // out row r <- in rows 2r + kh
    static constexpr int NTAPS = 4, NWT = 4, XR = 2 * TH + 2, XC = TW, CH = 0;
    static __device__ __forceinline__ int in_row0(int h0) { return 2 * h0; }
    static __device__ __forceinline__ int lrow(int tp, int wave) { return 2 * wave + tp; }
    static __device__ __forceinline__ int lcol(int) { return 0; }
    static __device__ __forceinline__ int wtap(int tp, int) { return tp; }
};
struct Up4 {        // out row r <- in rows (r - kh)/2, kh = (r&1) + 2j, j = 0,1   (tile row origin is even)
    static constexpr int NTAPS = 2, NWT = 4, XR = TH / 2 + 1, XC = TW, CH = 0;
    static __device__ __forceinline__ int in_row0(int h0) { return h0 / 2 - 1; }
    static __device__ __forceinline__ int lrow(int tp, int wave) { return (wave >> 1) - tp + 1; }
    static __device__ __forceinline__ int lcol(int) { return 0; }
    static __device__ __forceinline__ int wtap(int tp, int wave) { return (wave & 1) + 2 * tp; }
};

// DMA = true: the input tile is brought in by LDS-DMA (global_load_lds_dwordx4, 16 B per lane, 1 KiB per wave
// instruction, destination linear in LDS).  Rows then start 16-byte aligned at t0 - HL (HL = 4 when the geometry has a
// column halo) and are XCP = 64 (+8) floats wide, planes are exactly XR*XCP floats apart.  DMA = false: the register
// staged tile (any T, optional ELU' gating while staging) with the 17-mod-32 plane pitch.
template <int CIN, int COUT, class P, bool DMA, int PREC = 0>
struct Geo {
    static constexpr bool BF16 = PREC != 0;
    static constexpr int MT = (COUT + 15) / 16;
    static constexpr int CP = MT * 16;                       // weight image row pitch (floats)
    static constexpr bool SWZ = CP >= 32;
    // channels per staged chunk: DMA costs no registers, so mid-width layers move 8 channels per round trip
    // the (4,1) transposed geometry does 2 taps per channel: with 4-channel chunks a barrier interval holds too little
    // work, so it stages up to 16 channels per chunk (its tile is only TH/2 + 1 rows per channel)
    static constexpr int CC_STR = P::NTAPS == 2 ? (CIN < 16 ? CIN : 16) : (CIN < 8 ? CIN : (CIN <= 16 ? 8 : 4));   // measured
    static constexpr int CC = BF16 ? 4 : (P::NTAPS != 9 && DMA ? CC_STR : ((DMA && CIN >= 8 && CIN <= 16) ? 8 : 4));
    static constexpr int NCH = CIN / CC;
    static constexpr int HL = DMA ? (P::CH > 0 ? 4 : 0) : P::CH;
    static constexpr int XCP = DMA ? (P::CH > 0 ? TW + 8 : TW) : P::XC;
    // DMA layout: planes XR * XCP floats apart, padded by 16 when that is 0 mod 32 so that the two channel planes one
    // ds_read_b32 cycle touches (lane groups g, g + 1) fall into different bank halves
    static constexpr int PLANE = DMA ? P::XR * XCP + ((P::XR * XCP) % 32 == 0 ? 16 : 0) : plane_pad(P::XR * P::XC);
    static constexpr int ELEMS = CC * P::XR * P::XC;
    static constexpr int NLD = (ELEMS + NTHREADS - 1) / NTHREADS;
    static constexpr int NQ = DMA ? CC * PLANE / 4 : 0;      // 16-byte pieces of one chunk (DMA), plane pads included
    static constexpr int NPIECE = (NQ + 63) / 64;            // wave-wide DMA instructions per chunk
    static constexpr int BUF = DMA ? NPIECE * 256 : CC * PLANE;
    static constexpr int KW = P::NWT * CIN;                  // rows of the weight image
    // bf16 weight image of the 3x3 geometry: per 4-channel chunk two K=32 blocks (taps 0..7, then tap 8 + zeros),
    // [chunk][slot][co][8] bf16 with slot 0..3 = block 0 / lane group g and slot 4 = block 1 / lane group 0 (the other
    // groups of block 1 multiply zeros) -> one ds_read_b128 per A fragment.  Split mode appends the lo image.
    static constexpr int WBF_HALF = (CIN / 4) * 5 * CP * 8;             // bf16 elements of one image
    static constexpr int WBF_FLOATS = WBF_HALF / 2 * (PREC == 2 ? 2 : 1);
    static constexpr int W_FLOATS = BF16 ? WBF_FLOATS : KW * CP;
    // bf16 modes stage the tile as channel-interleaved bf16: Z[buffer][hi | lo][pixel][4 channels], pixel = row * ZCP + col
    // over rows h0 - D .. h0 + TH + D - 1 and columns t0 - 4 .. t0 + TW + 3 -> one ds_read_b64 per tap of a B fragment
    static constexpr int ZCP = TW + 8;
    static constexpr int ZPX = P::XR * ZCP;
    static constexpr int NIMG = PREC == 2 ? 2 : 1;
    static constexpr int Z_HALF = ZPX * 4;                    // bf16 elements of one image of one buffer
    static constexpr int Z_FLOATS = 2 * NIMG * Z_HALF / 2;
    static constexpr int ZPAIR = ZPX / 2;                     // pixel pairs staged per chunk (float2 per plane)
    static constexpr int ZLD = (ZPAIR + NTHREADS - 1) / NTHREADS;
    static constexpr int XS_FLOATS = BF16 ? Z_FLOATS : 2 * BUF;
};

__device__ float4 g_zero16;      // 16 zero bytes: DMA source of out-of-image pieces

__device__ __forceinline__ void glds16(const float* gsrc, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Workgroup w runs on XCD w % 8 (round-robin dispatch), and every XCD has its own L2.  Work item v of the persistent
// loop is therefore sent to tile (v % 8) * (ntiles / 8) + v / 8: each XCD sweeps one contiguous eighth of the raster,
// so the tiles that share halo rows and columns are in flight on the same L2 at about the same time.
__device__ __forceinline__ int xcd_tile(int v, int ntiles) {
    const int per = ntiles >> 3;
    return v < (per << 3) ? (v & 7) * per + (v >> 3) : v;
}

struct Tile { int b, h0, t0; };
__device__ __forceinline__ Tile decode_tile(int v, int tiles_h, int tiles_t, int ntiles) {
    Tile r;
    int tile = xcd_tile(v, ntiles);
    const int tt = tile % tiles_t; tile /= tiles_t;
    const int th = tile % tiles_h;
    r.b = tile / tiles_h; r.h0 = th * TH; r.t0 = tt * TW;
    return r;
}

// weight image  Wimg[(wt*CIN + c)*CP + swz(m)] = w[w_off + m*s_m + c*s_c + wt*s_t]   (0 for m >= COUT)
template <int CIN, int COUT, class P>
__device__ __forceinline__ void build_weight_image(float* img, const float* __restrict__ w, long s_m, long s_c, long s_t,
                                                   long w_off, int tid) {
    using G = Geo<CIN, COUT, P, false>;
    for (int i = tid; i < G::W_FLOATS; i += NTHREADS) {
        const int k = i / G::CP, m = i - k * G::CP;
        const int wt = k / CIN, c = k - wt * CIN;
        const int ms = G::SWZ ? (m ^ ((k & 1) << 4)) : m;
        img[k * G::CP + ms] = (m < COUT) ? w[w_off + m * s_m + c * s_c + wt * s_t] : 0.f;
    }
}

// bf16 image (3x3 geometry only): slot (chunk c4, slot sl, co, j) holds W(co, ci = 4 c4 + (j & 3), tap) with
// tap = 2 sl + (j >> 2) for sl < 4 and tap = 8 (j < 4 only) for sl = 4.  PREC == 2 appends the image of the residuals.
template <int CIN, int COUT, int PREC>
__device__ __forceinline__ void build_weight_image_bf16(__bf16* img, const float* __restrict__ w, long s_m, long s_c, long s_t,
                                                        long w_off, int tid) {
    constexpr int CP = ((COUT + 15) / 16) * 16;
    constexpr int TOTAL = (CIN / 4) * 5 * CP * 8;
    for (int i = tid; i < TOTAL; i += NTHREADS) {
        const int j = i & 7;
        int r = i >> 3;
        const int co = r % CP; r /= CP;
        const int sl = r % 5; const int c4 = r / 5;
        const int tap = sl < 4 ? 2 * sl + (j >> 2) : (j < 4 ? 8 : 9), ci = 4 * c4 + (j & 3);
        float v = 0.f;
        if (co < COUT && tap < 9) v = w[w_off + co * s_m + ci * s_c + tap * s_t];
        const __bf16 h = (__bf16)v;
        img[i] = h;
        if constexpr (PREC == 2) img[TOTAL + i] = (__bf16)(v - (float)h);
    }
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// Main loop of the bf16 / split-bf16 modes (3x3 geometry).  A chunk of 4 input channels is loaded 8 bytes per lane
// and plane into registers while the previous chunk is multiplied, converted ONCE per element to bf16 (hi, and lo in
// split mode) and written channel-interleaved into the other Z buffer: a B fragment (K = 2 taps x 4 channels per lane
// group) is then two ds_read_b64, with no conversion in the multiply loop.  One barrier per chunk.
template <int CIN, int COUT, class P, int PREC, class Epi>
__device__ __forceinline__ void conv_mainloop_z(const float* __restrict__ x, const float* Wimg, float* xs, int B, int H, int T,
                                                Epi&& epi) {
    using G = Geo<CIN, COUT, P, true, PREC>;
    constexpr int D = P::CH;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int tiles_h = (H + TH - 1) / TH, tiles_t = (T + TW - 1) / TW;
    const int ntiles = B * tiles_h * tiles_t;
    const long plane = (long)H * T;
    __bf16* Z = reinterpret_cast<__bf16*>(xs);
    const __bf16* wimg = reinterpret_cast<const __bf16*>(Wimg);

    float2 pre[G::ZLD][4];
    int p_row0 = 0, p_col0 = 0;
    auto issue = [&](int tile, int chunk) {
        const Tile tl = decode_tile(tile, tiles_h, tiles_t, ntiles);
        p_row0 = tl.h0 - D;
        p_col0 = tl.t0 - 4;
        const float* xb = x + ((long)tl.b * CIN + chunk * 4) * plane;
#pragma unroll
        for (int j = 0; j < G::ZLD; ++j) {
            int pp = tid + NTHREADS * j;
            if (pp >= G::ZPAIR) pp = G::ZPAIR - 1;
            const int r = pp / (G::ZCP / 2), c2 = pp - r * (G::ZCP / 2);
            int h = p_row0 + r, t = p_col0 + 2 * c2;
            h = h < 0 ? 0 : (h >= H ? H - 1 : h);
            t = t < 0 ? 0 : (t >= T ? T - 2 : t);
            const int o = h * T + t;
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) pre[j][ci] = *reinterpret_cast<const float2*>(xb + ci * plane + o);
        }
    };
    auto commit = [&](int buf) {
        __bf16* zh = Z + (buf * G::NIMG) * G::Z_HALF;
#pragma unroll
        for (int j = 0; j < G::ZLD; ++j) {
            const int pp = tid + NTHREADS * j;
            if (pp < G::ZPAIR) {
                const int r = pp / (G::ZCP / 2), c2 = pp - r * (G::ZCP / 2);
                const int h = p_row0 + r, t = p_col0 + 2 * c2;
                const bool ok = h >= 0 && h < H && t >= 0 && t < T;
                float v[8];
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) { v[ci] = ok ? pre[j][ci].x : 0.f; v[4 + ci] = ok ? pre[j][ci].y : 0.f; }
                bf16x8 hi, lo;
                split_bf16<PREC>(v, hi, lo);
                *reinterpret_cast<bf16x8*>(zh + pp * 8) = hi;
                if constexpr (PREC == 2) *reinterpret_cast<bf16x8*>(zh + G::Z_HALF + pp * 8) = lo;
            }
        }
    };

    // element offsets of this lane's taps inside a Z image (tap 2g, tap 2g+1, tap 8)
    const int ta = 2 * g, tb = 2 * g + 1;
    // column n = l15 of 16-pixel group nt is pixel 4 * l15 + nt of the row: a lane's four accumulators of one channel are
    // four CONSECUTIVE pixels, so the epilogue moves 16 bytes per lane
    const int pa = ((wave + (ta / 3) * D) * G::ZCP + (4 - D) + (ta % 3) * D + 4 * l15) * 4;
    const int pb = ((wave + (tb / 3) * D) * G::ZCP + (4 - D) + (tb % 3) * D + 4 * l15) * 4;
    const int p8 = ((wave + 2 * D) * G::ZCP + (4 - D) + 2 * D + 4 * l15) * 4;

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    int buf = 0;
    issue(tile, 0);
    commit(0);
    __syncthreads();
    for (; tile < ntiles; tile += gridDim.x) {
        f32x4 acc[G::MT][4];
#pragma unroll
        for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int chunk = 0; chunk < G::NCH; ++chunk) {
            const bool more_chunks = chunk + 1 < G::NCH;
            const bool more = more_chunks || tile + (int)gridDim.x < ntiles;
            if (more) issue(more_chunks ? tile : tile + (int)gridDim.x, more_chunks ? chunk + 1 : 0);
            const __bf16* zh = Z + (buf * G::NIMG) * G::Z_HALF;
            bf16x8 a0[G::MT], a1[G::MT], a0l[G::MT], a1l[G::MT];
            const bf16x8 zero8 = {};
#pragma unroll
            for (int mt = 0; mt < G::MT; ++mt) {
                const int o0 = (((chunk * 5 + g) * G::CP) + mt * 16 + l15) * 8, o1 = (((chunk * 5 + 4) * G::CP) + mt * 16 + l15) * 8;
                a0[mt] = *reinterpret_cast<const bf16x8*>(wimg + o0);
                a1[mt] = g == 0 ? *reinterpret_cast<const bf16x8*>(wimg + o1) : zero8;
                if constexpr (PREC == 2) {
                    a0l[mt] = *reinterpret_cast<const bf16x8*>(wimg + G::WBF_HALF + o0);
                    a1l[mt] = g == 0 ? *reinterpret_cast<const bf16x8*>(wimg + G::WBF_HALF + o1) : zero8;
                }
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                bf16x8 b0, b1, b0l, b1l;
                {
                    const uint2 ua = *reinterpret_cast<const uint2*>(zh + pa + nt * 4), ub = *reinterpret_cast<const uint2*>(zh + pb + nt * 4);
                    uint2 u8 = *reinterpret_cast<const uint2*>(zh + p8 + nt * 4);
                    if (g != 0) u8 = uint2{0u, 0u};
                    b0 = as_bf16x8(u32x4{ua.x, ua.y, ub.x, ub.y});
                    b1 = as_bf16x8(u32x4{u8.x, u8.y, 0u, 0u});
                }
                if constexpr (PREC == 2) {
                    const __bf16* zl = zh + G::Z_HALF;
                    const uint2 ua = *reinterpret_cast<const uint2*>(zl + pa + nt * 4), ub = *reinterpret_cast<const uint2*>(zl + pb + nt * 4);
                    uint2 u8 = *reinterpret_cast<const uint2*>(zl + p8 + nt * 4);
                    if (g != 0) u8 = uint2{0u, 0u};
                    b0l = as_bf16x8(u32x4{ua.x, ua.y, ub.x, ub.y});
                    b1l = as_bf16x8(u32x4{u8.x, u8.y, 0u, 0u});
                }
#pragma unroll
                for (int mt = 0; mt < G::MT; ++mt) {
                    acc[mt][nt] = mma_bf16<PREC>(a0[mt], a0l[mt], b0, b0l, acc[mt][nt]);
                    acc[mt][nt] = mma_bf16<PREC>(a1[mt], a1l[mt], b1, b1l, acc[mt][nt]);
                }
            }
            if (!more_chunks) epi(decode_tile(tile, tiles_h, tiles_t, ntiles), acc);
            if (more) commit(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }
}


// The pipelined implicit-GEMM main loop over the tiles of one workgroup.  `epi(tile, acc)` consumes a finished tile.
struct NoChunkHook {
    __device__ __forceinline__ void operator()(int, const float*) const {}
};

// `hook(chunk, tile)` runs once per staged chunk, after the barrier that makes the chunk's LDS tile visible and before the
// next DMA may overwrite the other buffer: the fused residual block uses it to pick its residual values out of the tile.
template <int CIN, int COUT, class P, bool GATE, bool DMA, int PREC, class Epi, class Hook = NoChunkHook>
__device__ __forceinline__ void conv_mainloop(const float* __restrict__ x, const float* __restrict__ gy, const float* Wimg,
                                              float* xs, int B, int Hin, int Hout, int T, Epi&& epi, Hook&& hook = Hook()) {
    static_assert(!(GATE && DMA), "gated staging needs the register path");
    constexpr bool BF16 = PREC != 0;
    static_assert(!BF16 || (DMA && P::NTAPS == 9), "bf16 operands: 3x3 geometry on the DMA path");
    using G = Geo<CIN, COUT, P, DMA, PREC>;
    if constexpr (BF16) {
        conv_mainloop_z<CIN, COUT, P, PREC>(x, Wimg, xs, B, Hin, T, epi);
        return;
    }
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int tiles_h = (Hout + TH - 1) / TH, tiles_t = (T + TW - 1) / TW;
    const int ntiles = B * tiles_h * tiles_t;
    const long plane = (long)Hin * T;
    int acol[G::MT];
#pragma unroll
    for (int mt = 0; mt < G::MT; ++mt) acol[mt] = G::SWZ ? ((mt * 16 + l15) ^ ((g & 1) << 4)) : (mt * 16 + l15);

    // ---- register staging (DMA == false): unconditional loads from a clamped address, zeroed at commit ----
    float pre[DMA ? 1 : G::NLD];
    float preg[GATE ? G::NLD : 1];
    int p_row0 = 0, p_col0 = 0;
    auto issue_reg = [&](int tile, int chunk) {
        const Tile tl = decode_tile(tile, tiles_h, tiles_t, ntiles);
        p_row0 = P::in_row0(tl.h0);
        p_col0 = tl.t0 - P::CH;
        const float* xb = x + ((long)tl.b * CIN + chunk * G::CC) * plane;
        const float* gb = GATE ? gy + ((long)tl.b * CIN + chunk * G::CC) * plane : nullptr;
#pragma unroll
        for (int j = 0; j < (DMA ? 0 : G::NLD); ++j) {
            int e = tid + NTHREADS * j;
            if (e >= G::ELEMS) e = G::ELEMS - 1;
            const int ci = e / (P::XR * P::XC);
            const int rem = e - ci * (P::XR * P::XC);
            const int r = rem / P::XC, c = rem - r * P::XC;
            int h = p_row0 + r, t = p_col0 + c;
            h = h < 0 ? 0 : (h >= Hin ? Hin - 1 : h);
            t = t < 0 ? 0 : (t >= T ? T - 1 : t);
            const int o = ci * (int)plane + h * T + t;
            pre[j] = xb[o];
            if (GATE) preg[j] = gb[o];
        }
    };
    auto commit_reg = [&](int buf) {
        float* dst = xs + buf * G::BUF;
#pragma unroll
        for (int j = 0; j < (DMA ? 0 : G::NLD); ++j) {
            const int e = tid + NTHREADS * j;
            if (e < G::ELEMS) {
                const int ci = e / (P::XR * P::XC);
                const int rem = e - ci * (P::XR * P::XC);
                const int r = rem / P::XC, c = rem - r * P::XC;
                const int h = p_row0 + r, t = p_col0 + c;
                float v = pre[j];
                if (GATE) v *= elu_grad_from_out(preg[j]);
                dst[ci * G::PLANE + rem] = (h >= 0 && h < Hin && t >= 0 && t < T) ? v : 0.f;
            }
        }
    };
    // ---- LDS-DMA staging (DMA == true): each wave moves NPIECE/8 one-KiB pieces straight into the other buffer ----
    auto issue_dma = [&](int tile, int chunk, int buf) {
        const Tile tl = decode_tile(tile, tiles_h, tiles_t, ntiles);
        const int row0 = P::in_row0(tl.h0), col0 = tl.t0 - G::HL;
        const float* xb = x + ((long)tl.b * CIN + chunk * G::CC) * plane;
        float* dst = xs + buf * G::BUF;
        constexpr int RQ = G::XCP / 4;              // pieces per row
        constexpr int PQ = G::PLANE / 4;            // pieces per (padded) channel plane
#pragma unroll
        for (int jj = 0; jj < (G::NPIECE + 7) / 8; ++jj) {
            const int j = __builtin_amdgcn_readfirstlane(wave) + 8 * jj;      // provably wave-uniform LDS base
            if (j < G::NPIECE) {
                const int q = j * 64 + lane;
                const int ci = q / PQ;
                const int rem = q - ci * PQ;
                const int r = rem / RQ, c4 = rem - r * RQ;
                const int h = row0 + r, t = col0 + 4 * c4;
                const bool ok = q < G::NQ && r < P::XR && h >= 0 && h < Hin && t >= 0 && t < T;
                const float* src = ok ? xb + (ci * (int)plane + h * T + t) : reinterpret_cast<const float*>(&g_zero16);
                glds16(src, dst + j * 256);
            }
        }
    };

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    int buf = 0;
    if (DMA) issue_dma(tile, 0, 0); else issue_reg(tile, 0);
    for (; tile < ntiles; tile += gridDim.x) {
        f32x4 acc[G::MT][4];
#pragma unroll
        for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int chunk = 0; chunk < G::NCH; ++chunk) {
            if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces have landed
            else commit_reg(buf);
            __syncthreads();
            // prefetch the next work item while this one is multiplied
            const bool more_chunks = chunk + 1 < G::NCH;
            const bool more_tiles = tile + (int)gridDim.x < ntiles;
            if (more_chunks || more_tiles) {
                const int nt_ = more_chunks ? tile : tile + (int)gridDim.x;
                const int nc_ = more_chunks ? chunk + 1 : 0;
                if (DMA) issue_dma(nt_, nc_, buf ^ 1); else issue_reg(nt_, nc_);
            }
            const float* xb = xs + buf * G::BUF;
            const int c0 = chunk * G::CC;
            hook(chunk, xb);
            {
#pragma unroll
            for (int tp = 0; tp < P::NTAPS; ++tp) {
                const int wt = P::wtap(tp, wave);
                const float* bp0 = xb + g * G::PLANE + P::lrow(tp, wave) * G::XCP + (G::HL - P::CH) + P::lcol(tp) + l15;
                const float* ap0 = Wimg + (wt * CIN + c0 + g) * G::CP;
#pragma unroll
                for (int cc = 0; cc < G::CC; cc += 4) {
                    float a[G::MT];
#pragma unroll
                    for (int mt = 0; mt < G::MT; ++mt) a[mt] = ap0[cc * G::CP + acol[mt]];
                    const float* bp = bp0 + cc * G::PLANE;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const float bv = bp[nt * 16];
#pragma unroll
                        for (int mt = 0; mt < G::MT; ++mt) acc[mt][nt] = mfma16(a[mt], bv, acc[mt][nt]);
                    }
                }
            }
            }
            buf ^= 1;
        }
        epi(decode_tile(tile, tiles_h, tiles_t, ntiles), acc);
    }
}

// ---- plain convolution kernel: out = act(conv + bias) + res -----------------------------------------------
struct WSpec { long s_m, s_c, s_t, off; };

template <int CIN, int COUT, class P, bool GATE, bool DMA, int PREC>
__global__ __launch_bounds__(NTHREADS, 4) void k_conv_mfma(const float* __restrict__ x, const float* __restrict__ gy,
                                                        const float* __restrict__ w, WSpec ws, const float* __restrict__ bias,
                                                        const float* __restrict__ res, float* __restrict__ y, int B, int Hin,
                                                        int Hout, int T, int act) {
    using G = Geo<CIN, COUT, P, DMA, PREC>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;                       // DMA destinations first: 16-byte aligned
    float* Wimg = lds + G::XS_FLOATS;
    if constexpr (PREC != 0) build_weight_image_bf16<CIN, COUT, PREC>(reinterpret_cast<__bf16*>(Wimg), w, ws.s_m, ws.s_c, ws.s_t, ws.off, threadIdx.x);
    else build_weight_image<CIN, COUT, P>(Wimg, w, ws.s_m, ws.s_c, ws.s_t, ws.off, threadIdx.x);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, l15 = lane & 15;
    const long oplane = (long)Hout * T;
    conv_mainloop<CIN, COUT, P, GATE, DMA, PREC>(x, gy, Wimg, xs, B, Hin, Hout, T, [&](const Tile& tl, f32x4 (&acc)[G::MT][4]) {
        // addressing: a wave-uniform 64-bit base per (clip, channel mt*16 + r) plus one 32-bit lane offset per 16-column
        // group (channel part 4g of the lane included) -- no per-element 64-bit pointers to keep alive
        const int ub = __builtin_amdgcn_readfirstlane(tl.b), uh0 = __builtin_amdgcn_readfirstlane(tl.h0),
                  ut0 = __builtin_amdgcn_readfirstlane(tl.t0);
        const int h = uh0 + wave;
        if (h >= Hout) return;
        static_assert(COUT % 4 == 0, "lane groups own 4 consecutive output channels");
        const int g4 = (COUT % 16 == 0) ? 4 * g : (4 * g < COUT ? 4 * g : COUT - 4);      // clamped: loads stay in bounds
        const long cbase = (long)ub * COUT * oplane;
        if constexpr (PREC != 0 || DMA) {
            // 16-byte loads and stores: in the bf16 main loop accumulators nt = 0..3 of a lane already are pixels t .. t + 3;
            // the fp32 loop (pixel nt * 16 + l15) gets there with a quad transpose of the finished values
            const int t = PREC != 0 ? ut0 + 4 * l15 : ut0 + 16 * (l15 & 3) + 4 * (l15 >> 2);
            const bool tv = t < T;
            const unsigned vo = (unsigned)(g4 * (int)oplane + h * T + (tv ? t : T - 4));
            float4 rv[G::MT][4];
            if (res) {
#pragma unroll
                for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        rv[mt][r] = *reinterpret_cast<const float4*>(res + cbase + (long)((mt * 16 + r < COUT) ? mt * 16 + r : 0) * oplane + vo);
            }
#pragma unroll
            for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mt * 16 + 4 * g + r;
                    const float bv = (bias && m < COUT) ? bias[m] : 0.f;
                    float v[4];
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        v[nt] = acc[mt][nt][r] + bv;
                        if (act == TT_ACT_ELU) v[nt] = elu1(v[nt]);
                    }
                    if constexpr (PREC == 0) quad_transpose(v, lane);
                    if (m >= COUT || !tv) continue;
                    if (res) { v[0] += rv[mt][r].x; v[1] += rv[mt][r].y; v[2] += rv[mt][r].z; v[3] += rv[mt][r].w; }
                    *reinterpret_cast<float4*>(y + cbase + (long)(mt * 16 + r) * oplane + vo) = float4{v[0], v[1], v[2], v[3]};
                }
            return;
        }
        unsigned vo[4];
        bool tv[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int t = ut0 + nt * 16 + l15;
            tv[nt] = t < T;
            vo[nt] = (unsigned)(g4 * (int)oplane + h * T + (tv[nt] ? t : T - 1));
        }
        float rv[G::MT][4][4];
        if (res) {      // all residual loads first (branch-free): one latency instead of one per store
#pragma unroll
            for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int mu = (mt * 16 + r < COUT) ? mt * 16 + r : 0;
                    const float* rb = res + cbase + (long)mu * oplane;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) rv[mt][r][nt] = rb[vo[nt]];
                }
        }
#pragma unroll
        for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mt * 16 + 4 * g + r;
                if (m >= COUT) continue;
                const float bv = bias ? bias[m] : 0.f;
                float* yb = y + cbase + (long)(mt * 16 + r) * oplane;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    if (!tv[nt]) continue;
                    float v = acc[mt][nt][r] + bv;
                    if (act == TT_ACT_ELU) v = elu1(v);
                    if (res) v += rv[mt][r][nt];
                    yb[vo[nt]] = v;
                }
            }
    });
}

// ---- fused residual block -------------------------------------------------------------------------------------
template <int C>
struct RB {
    static constexpr int MT = (C + 15) / 16, CPAD = MT * 16, CP = CPAD;
    static constexpr bool SWZ = CP >= 32;
    static constexpr int TP = 17;
    static constexpr int TR_FLOATS = 8 * 2 * CPAD * TP;      // 8 waves x {dA2, h1} x CPAD x 16 pixels
    // W2 images: W2s[c_in][co2] (forward 1x1) and W2t[co2][c_in] (its transpose), both swizzled like Wimg
    // rows used together by the four lane groups differ by 4: swizzle the 16-column halves by bit 2 of the row
    static __device__ __forceinline__ int swz(int row, int col) { return SWZ ? (col ^ (((row >> 2) & 1) << 4)) : col; }
};

template <int C>
__device__ __forceinline__ void build_w2_images(float* W2s, float* W2t, float* b1s, float* b2s, const float* __restrict__ w2,
                                                const float* __restrict__ b1, const float* __restrict__ b2, int tid) {
    using R = RB<C>;
    for (int i = tid; i < R::CPAD * R::CP; i += NTHREADS) {
        const int r = i / R::CP, c = i - r * R::CP;
        const bool ok = r < C && c < C;
        W2s[r * R::CP + R::swz(r, c)] = ok ? w2[c * C + r] : 0.f;       // row = c_in, col = co2
        if (W2t) W2t[r * R::CP + R::swz(r, c)] = ok ? w2[r * C + c] : 0.f;   // row = co2, col = c_in
    }
    for (int i = tid; i < R::CPAD; i += NTHREADS) {
        b1s[i] = i < C ? b1[i] : 0.f;
        b2s[i] = i < C ? b2[i] : 0.f;
    }
}

template <int C, int D, bool DMA, int PREC>
__global__ __launch_bounds__(NTHREADS, 4) void k_rb_fwd(const float* __restrict__ x, const float* __restrict__ w1,
                                                     const float* __restrict__ b1, const float* __restrict__ w2,
                                                     const float* __restrict__ b2, float* __restrict__ y,
                                                     float* __restrict__ h1out, int B, int H, int T) {
    using P = Res3x3<D>;
    using G = Geo<C, C, P, DMA, PREC>;
    using R = RB<C>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;
    float* Wimg = xs + G::XS_FLOATS;
    float* W2s = Wimg + G::W_FLOATS;
    float* b1s = W2s + R::CPAD * R::CP;
    float* b2s = b1s + R::CPAD;
    if constexpr (PREC != 0) build_weight_image_bf16<C, C, PREC>(reinterpret_cast<__bf16*>(Wimg), w1, (long)C * 9, 9, 1, 0, threadIdx.x);
    else build_weight_image<C, C, P>(Wimg, w1, (long)C * 9, 9, 1, 0, threadIdx.x);
    build_w2_images<C>(W2s, nullptr, b1s, b2s, w2, b1, b2, threadIdx.x);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, l15 = lane & 15;
    const long plane = (long)H * T;
    // fp32 path on aligned tensors: the residual x never has to be read again from memory -- every channel passes through the
    // LDS tile during the main loop, and the lane group that owns a chunk's channels in the epilogue (channels 4 g + r of 16-row
    // tile mt: chunk j = 4 mt + g at 4 channels per chunk) copies its four consecutive centre pixels out with one ds_read_b128
    // per channel while the chunk is resident.  32 registers at C = 32 for 0.55 GB less traffic per launch.
    constexpr bool RES_LDS = TT_RES_LDS && PREC == 0 && DMA && (G::CC == 4 || G::CC == 8) && (C % 16 == 0);
    float4 xkeep[RES_LDS ? G::MT : 1][4];
    auto keep_residual = [&](int chunk, const float* tile) {
        if constexpr (RES_LDS) {
            const float* p = tile + (wave + D) * G::XCP + G::HL + 16 * (l15 & 3) + 4 * (l15 >> 2);
#pragma unroll
            for (int mt = 0; mt < G::MT; ++mt) {
                const int first = 16 * mt + 4 * g;                    // first of this lane's four channels in row tile mt
                if (chunk == first / G::CC) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) xkeep[mt][r] = *reinterpret_cast<const float4*>(p + (first % G::CC + r) * G::PLANE);
                }
            }
        }
    };
    conv_mainloop<C, C, P, false, DMA, PREC>(x, nullptr, Wimg, xs, B, H, H, T, [&](const Tile& tl, f32x4 (&acc)[G::MT][4]) {
        // addressing as in k_conv_mfma: uniform base per (clip, channel m2*16 + r) + 32-bit lane offsets; in the bf16 modes
        // (PX4) accumulators nt = 0..3 are four consecutive pixels: 16 bytes per lane for the residual, h1 and y
        constexpr bool PX4 = PREC != 0;            // lane owns pixels 4 l15 + nt
        constexpr bool QT = PREC == 0 && DMA;      // fp32 loop on aligned tensors: quad transpose, then the same 16-byte accesses
        const int ub = __builtin_amdgcn_readfirstlane(tl.b), uh0 = __builtin_amdgcn_readfirstlane(tl.h0),
                  ut0 = __builtin_amdgcn_readfirstlane(tl.t0);
        const int h = uh0 + wave;
        const int g4 = (C % 16 == 0) ? 4 * g : (4 * g < C ? 4 * g : C - 4);
        unsigned vo[4];
        bool tv[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int t = PX4 ? ut0 + 4 * l15 + nt : (QT ? ut0 + 16 * (l15 & 3) + 4 * (l15 >> 2) + nt : ut0 + nt * 16 + l15);
            tv[nt] = t < T && h < H;
            vo[nt] = (unsigned)(g4 * (int)plane + (h < H ? h : H - 1) * T + (t < T ? t : T - ((PX4 || QT) ? 4 - nt : 1)));
        }
        const long cbase = (long)ub * C * plane;
        f32x4 acc2[G::MT][4];
#pragma unroll
        for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                acc2[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mt][nt][r] = elu1(acc[mt][nt][r] + b1s[mt * 16 + 4 * g + r]);
            }
        if (h1out) {       // hidden activation for the backward pass
#pragma unroll
            for (int m2 = 0; m2 < G::MT; ++m2)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* hb = h1out + cbase + (long)(m2 * 16 + r) * plane;
                    if constexpr (PX4 || QT) {
                        float hv[4] = {acc[m2][0][r], acc[m2][1][r], acc[m2][2][r], acc[m2][3][r]};
                        if constexpr (QT) quad_transpose(hv, lane);
                        if (m2 * 16 + 4 * g + r < C && tv[0]) *reinterpret_cast<float4*>(hb + vo[0]) = float4{hv[0], hv[1], hv[2], hv[3]};
                    } else {
                        if (m2 * 16 + 4 * g + r >= C) continue;
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
                            if (tv[nt]) hb[vo[nt]] = acc[m2][nt][r];
                    }
                }
        }
#pragma unroll
        for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kr = mt * 16 + 4 * g + r;        // k index of this lane group for step (mt, r)
                float a2[G::MT];
#pragma unroll
                for (int m2 = 0; m2 < G::MT; ++m2) a2[m2] = W2s[kr * R::CP + R::swz(kr, m2 * 16 + l15)];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int m2 = 0; m2 < G::MT; ++m2) acc2[m2][nt] = mfma16(a2[m2], acc[mt][nt][r], acc2[m2][nt]);
            }
        if (h >= H) return;
        // the residual input: all loads issued together once the 3x3 accumulators are dead (one exposed latency per tile)
        float xres[G::MT][4][4];
#pragma unroll
        for (int m2 = 0; m2 < G::MT; ++m2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (RES_LDS) {
                    xres[m2][r][0] = xkeep[m2][r].x; xres[m2][r][1] = xkeep[m2][r].y; xres[m2][r][2] = xkeep[m2][r].z; xres[m2][r][3] = xkeep[m2][r].w;
                    continue;
                }
                const float* xb = x + cbase + (long)((m2 * 16 + r < C) ? m2 * 16 + r : 0) * plane;
                if constexpr (PX4 || QT) {
                    const float4 v = *reinterpret_cast<const float4*>(xb + vo[0]);
                    xres[m2][r][0] = v.x; xres[m2][r][1] = v.y; xres[m2][r][2] = v.z; xres[m2][r][3] = v.w;
                } else {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) xres[m2][r][nt] = xb[vo[nt]];
                }
            }
#pragma unroll
        for (int m2 = 0; m2 < G::MT; ++m2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = m2 * 16 + 4 * g + r;
                const float bias = b2s[co < C ? co : 0];
                float* yb = y + cbase + (long)(m2 * 16 + r) * plane;
                if constexpr (PX4 || QT) {
                    float ov[4] = {elu1(acc2[m2][0][r] + bias), elu1(acc2[m2][1][r] + bias), elu1(acc2[m2][2][r] + bias),
                                   elu1(acc2[m2][3][r] + bias)};
                    if constexpr (QT) quad_transpose(ov, lane);
                    if (co < C && tv[0]) *reinterpret_cast<float4*>(yb + vo[0]) =
                        float4{ov[0] + xres[m2][r][0], ov[1] + xres[m2][r][1], ov[2] + xres[m2][r][2], ov[3] + xres[m2][r][3]};
                } else {
                    if (co >= C) continue;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        if (tv[nt]) yb[vo[nt]] = elu1(acc2[m2][nt][r] + bias) + xres[m2][r][nt];
                }
            }
    }, keep_residual);
}

__device__ __forceinline__ float group16_sum(float v) {
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
    return v;
}

// wave-local LDS hand-off: LDS operations of one wave execute in program order, so only the compiler
// must be kept from reordering and the writes must have landed before other lanes read them.
__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// RECOMP = true : hidden activation recomputed from x (3x3 conv main loop);  false : read back from `h1in`
// NT = 16-pixel column groups a wave holds at a time (a tile is always 8 rows x 64 columns).  The fragments of h1, dA2, dH1 and
// the dW2 / bias accumulators are all live together: at C = 32 with NT = 4 that is 256 registers -- ONE workgroup per CU, whose
// loads, matrix chain and stores then run back to back.  NT = 2 (two passes per tile, 8-byte instead of 16-byte accesses) was
// written to get under 128 registers and two workgroups per CU; it still spills and is slower (launch_rb_bwd_a_v), so NT = 4 ships.
template <int C, int D, bool DMA, bool RECOMP, int NT = 4>
__global__ __launch_bounds__(NTHREADS, (NT == 2 ? 4 : 1)) void k_rb_bwd_a(const float* __restrict__ x, const float* __restrict__ h1in,
                                                       const float* __restrict__ dy,
                                                       const float* __restrict__ w1, const float* __restrict__ b1,
                                                       const float* __restrict__ w2, const float* __restrict__ b2,
                                                       float* __restrict__ da1, float* __restrict__ db1,
                                                       float* __restrict__ dw2, float* __restrict__ db2, int B, int H, int T) {
    using P = Res3x3<D>;
    using G = Geo<C, C, P, DMA>;
    using R = RB<C>;
    static_assert(NT == 4 || (NT == 2 && !RECOMP), "the recompute path takes whole 64-column tiles from the conv main loop");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;
    float* Wimg = xs + (RECOMP ? G::XS_FLOATS : 0);
    float* W2s = Wimg + (RECOMP ? G::W_FLOATS : 0);
    float* W2t = W2s + R::CPAD * R::CP;
    float* b1s = W2t + R::CPAD * R::CP;
    float* b2s = b1s + R::CPAD;
    float* tr = b2s + R::CPAD;          // per-wave transpose tiles (own region: no workgroup barrier needed)
    if (RECOMP) build_weight_image<C, C, P>(Wimg, w1, (long)C * 9, 9, 1, 0, threadIdx.x);
    build_w2_images<C>(W2s, W2t, b1s, b2s, w2, b1, b2, threadIdx.x);
    if (!RECOMP) __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, l15 = lane & 15;
    const long plane = (long)H * T;
    float* trA = tr + wave * 2 * R::CPAD * R::TP;
    float* trB = trA + R::CPAD * R::TP;

    f32x4 accw2[G::MT][G::MT];
    float db1acc[G::MT][4], db2acc[G::MT][4];
#pragma unroll
    for (int a = 0; a < G::MT; ++a) {
#pragma unroll
        for (int c = 0; c < G::MT; ++c) accw2[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) { db1acc[a][r] = 0.f; db2acc[a][r] = 0.f; }
    }

    // hidden activation read back (not recomputed): the pixel <-> (nt, lane) assignment is free, so a lane takes four
    // consecutive pixels per channel and every global access of the kernel is 16 bytes per lane
    const bool wide = !RECOMP && (C % 16 == 0) && (T % 4 == 0) && ((reinterpret_cast<uintptr_t>(h1in) | reinterpret_cast<uintptr_t>(dy) |
                                                                     reinterpret_cast<uintptr_t>(da1)) & 15) == 0;
    auto epi = [&](const Tile& tl, f32x4 (&h1)[G::MT][NT]) {       // tl.t0 = first column of the NT * 16 columns in hand
        const int h = tl.h0 + wave;
        f32x4 a2[G::MT][NT];
#pragma unroll
        for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                a2[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (RECOMP) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) h1[mt][nt][r] = elu1(h1[mt][nt][r] + b1s[mt * 16 + 4 * g + r]);
                }
            }
#pragma unroll
        for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kr = mt * 16 + 4 * g + r;
                float av[G::MT];
#pragma unroll
                for (int m2 = 0; m2 < G::MT; ++m2) av[m2] = W2s[kr * R::CP + R::swz(kr, m2 * 16 + l15)];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int m2 = 0; m2 < G::MT; ++m2) a2[m2][nt] = mfma16(av[m2], h1[mt][nt][r], a2[m2][nt]);
            }
        // dA2 = dy * ELU'(a2 + b2), in place
#pragma unroll
        for (int m2 = 0; m2 < G::MT; ++m2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = m2 * 16 + 4 * g + r;
                const float bias = b2s[co];
                const long base = ((long)tl.b * C + (co < C ? co : 0)) * plane + (long)(h < H ? h : 0) * T;
                float dv[NT];
                if (wide) {
                    const int t = tl.t0 + NT * l15;
                    if constexpr (NT == 4) {
                        float4 v = *reinterpret_cast<const float4*>(dy + base + (t < T ? t : 0));
                        if (!(h < H && t < T)) v = float4{0.f, 0.f, 0.f, 0.f};
                        dv[0] = v.x; dv[1] = v.y; dv[2] = v.z; dv[3] = v.w;
                    } else {
                        float2 v = *reinterpret_cast<const float2*>(dy + base + (t < T ? t : 0));
                        if (!(h < H && t < T)) v = float2{0.f, 0.f};
                        dv[0] = v.x; dv[1] = v.y;
                    }
                } else {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int t = tl.t0 + nt * 16 + l15;
                        dv[nt] = (co < C && h < H && t < T) ? dy[base + t] : 0.f;
                    }
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float gd = dv[nt] * elu_grad_from_out(elu1(a2[m2][nt][r] + bias));
                    a2[m2][nt][r] = gd;
                    db2acc[m2][r] += gd;
                }
            }
        // dH1 = W2^T . dA2 (dA2 fragments as B operands); dA1 = dH1 * ELU'(h1)
        f32x4 d1[G::MT][NT];
#pragma unroll
        for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) d1[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m2 = 0; m2 < G::MT; ++m2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kr = m2 * 16 + 4 * g + r;
                float av[G::MT];
#pragma unroll
                for (int mt = 0; mt < G::MT; ++mt) av[mt] = W2t[kr * R::CP + R::swz(kr, mt * 16 + l15)];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < G::MT; ++mt) d1[mt][nt] = mfma16(av[mt], a2[m2][nt][r], d1[mt][nt]);
            }
#pragma unroll
        for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = mt * 16 + 4 * g + r;
                const long base = ((long)tl.b * C + (co < C ? co : 0)) * plane + (long)(h < H ? h : 0) * T;
                float gv[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    gv[nt] = d1[mt][nt][r] * elu_grad_from_out(h1[mt][nt][r]);
                    db1acc[mt][r] += gv[nt];
                }
                if (wide) {
                    const int t = tl.t0 + NT * l15;
                    if constexpr (NT == 4) {
                        if (h < H && t < T) *reinterpret_cast<float4*>(da1 + base + t) = float4{gv[0], gv[1], gv[2], gv[3]};
                    } else {
                        if (h < H && t < T) *reinterpret_cast<float2*>(da1 + base + t) = float2{gv[0], gv[1]};
                    }
                } else {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int t = tl.t0 + nt * 16 + l15;
                        if (co < C && h < H && t < T) da1[base + t] = gv[nt];
                    }
                }
            }
        // dW2[co2][c] += sum_pix dA2[co2][pix] * h1[c][pix]: 16 pixels at a time through the wave's own LDS tiles
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            wave_lds_sync();
#pragma unroll
            for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    trA[(mt * 16 + 4 * g + r) * R::TP + l15] = a2[mt][nt][r];
                    trB[(mt * 16 + 4 * g + r) * R::TP + l15] = h1[mt][nt][r];
                }
            wave_lds_sync();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                float av[G::MT], bv[G::MT];
#pragma unroll
                for (int mt = 0; mt < G::MT; ++mt) {
                    av[mt] = trA[(mt * 16 + l15) * R::TP + ks * 4 + g];
                    bv[mt] = trB[(mt * 16 + l15) * R::TP + ks * 4 + g];
                }
#pragma unroll
                for (int m2 = 0; m2 < G::MT; ++m2)
#pragma unroll
                    for (int mt = 0; mt < G::MT; ++mt) accw2[m2][mt] = mfma16(av[m2], bv[mt], accw2[m2][mt]);
            }
        }
    };
    if constexpr (RECOMP) {
        conv_mainloop<C, C, P, false, DMA, 0>(x, nullptr, Wimg, xs, B, H, H, T, epi);
    } else {
        const int tiles_h = (H + TH - 1) / TH, tiles_t = (T + TW - 1) / TW;
        const int ntiles = B * tiles_h * tiles_t;
#pragma unroll 1
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            const Tile tl0 = decode_tile(tile, tiles_h, tiles_t, ntiles);
            const int h = tl0.h0 + wave;
#pragma unroll 1
            for (int sub = 0; sub < 4 / NT; ++sub) {
            Tile tl = tl0;
            tl.t0 = tl0.t0 + sub * 16 * NT;
            f32x4 h1[G::MT][NT];
            if (wide) {     // column n of group nt = pixel NT n + nt: a lane's NT values per channel are one 16- / 8-byte load
                const int t = tl.t0 + NT * l15;
                const bool ok = h < H && t < T;
                const long cb = (long)tl.b * C * plane;
                const unsigned vo = (unsigned)(4 * g * (int)plane + (h < H ? h : 0) * T + (t < T ? t : 0));
#pragma unroll
                for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if constexpr (NT == 4) {
                            float4 v = *reinterpret_cast<const float4*>(h1in + cb + (long)(mt * 16 + r) * plane + vo);
                            if (!ok) v = float4{0.f, 0.f, 0.f, 0.f};
                            h1[mt][0][r] = v.x; h1[mt][1][r] = v.y; h1[mt][2][r] = v.z; h1[mt][3][r] = v.w;
                        } else {
                            float2 v = *reinterpret_cast<const float2*>(h1in + cb + (long)(mt * 16 + r) * plane + vo);
                            if (!ok) v = float2{0.f, 0.f};
                            h1[mt][0][r] = v.x; h1[mt][1][r] = v.y;
                        }
                    }
            } else {
#pragma unroll
            for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = mt * 16 + 4 * g + r;
                    const long base = ((long)tl.b * C + (co < C ? co : 0)) * plane + (long)(h < H ? h : 0) * T;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int t = tl.t0 + nt * 16 + l15;
                        h1[mt][nt][r] = (co < C && h < H && t < T) ? h1in[base + t] : 0.f;
                    }
                }
            }
            epi(tl, h1);
            }
        }
    }
    // reduce the 8 waves in LDS (own region: tr), then one global atomic per element per workgroup
    __syncthreads();
    float* red = tr;                                   // [CPAD][CPAD] dW2, then db1[CPAD], db2[CPAD]
    for (int i = threadIdx.x; i < R::CPAD * R::CPAD + 2 * R::CPAD; i += NTHREADS) red[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int m2 = 0; m2 < G::MT; ++m2)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = m2 * 16 + 4 * g + r;
            const float s1 = group16_sum(db1acc[m2][r]), s2 = group16_sum(db2acc[m2][r]);
            if (l15 == 0) { atomicAdd(&red[R::CPAD * R::CPAD + co], s1); atomicAdd(&red[R::CPAD * R::CPAD + R::CPAD + co], s2); }
#pragma unroll
            for (int mt = 0; mt < G::MT; ++mt) atomicAdd(&red[co * R::CPAD + mt * 16 + l15], accw2[m2][mt][r]);
        }
    __syncthreads();
    for (int i = threadIdx.x; i < R::CPAD * R::CPAD; i += NTHREADS) {
        const int co = i / R::CPAD, c = i - co * R::CPAD;
        if (co < C && c < C) atomicAdd(dw2 + co * C + c, red[i]);
    }
    if (threadIdx.x < C) {
        atomicAdd(db1 + threadIdx.x, red[R::CPAD * R::CPAD + threadIdx.x]);
        atomicAdd(db2 + threadIdx.x, red[R::CPAD * R::CPAD + R::CPAD + threadIdx.x]);
    }
}

// ---- weight gradients as MFMA GEMMs with K = pixels ---------------------------------------------------------------
//   dW[a][b][tap] += sum_{r,t} P[a][r][t] * Q[b][qrow(r,tap)][t + qcol(tap)]          a < CA, b < CB
// grid.y splits b into NS slices of CBS channels; columns n = tap*CBS + bl.
// WTH rows (= waves) x WTW columns of P per tile: small channel counts are latency-bound per tile, so they get
// the big tile; wide layers keep LDS small enough for two workgroups per CU.
template <int D, int WTH_>
struct WRes {       // 3x3 dilated: Q rows r + kh*D - D, cols t + kw*D - D
    static constexpr int NTAPS = 9, WTH = WTH_, WTW = 64, XR = WTH + 2 * D, XC = WTW + 2 * D, CH = D;
    static __device__ __forceinline__ int q_row0(int h0) { return h0 - D; }
    static __device__ __forceinline__ int qoff(int tap) { return (tap / 3) * D * XC + (tap % 3) * D; }
    static __device__ __forceinline__ int qrow_of_wave(int wave) { return wave; }
};
template <int WTH_>
struct WStr {       // (4,1) stride 2: Q rows 2r + kh
    static constexpr int NTAPS = 4, WTH = WTH_, WTW = 64, XR = 2 * WTH + 2, XC = WTW, CH = 0;
    static __device__ __forceinline__ int q_row0(int h0) { return 2 * h0; }
    static __device__ __forceinline__ int qoff(int tap) { return tap * XC; }
    static __device__ __forceinline__ int qrow_of_wave(int wave) { return 2 * wave; }
};

template <int CA, int CBS, class WP>
struct WGeo {
    static constexpr int MT = (CA + 15) / 16, CAP = MT * 16;
    static constexpr int NN = WP::NTAPS * CBS, NTN = (NN + 15) / 16;
    static constexpr int PLANE = plane_pad(WP::XR * WP::XC);
    static constexpr int AP = WP::WTW + 1;
    static constexpr int Q_FLOATS = CBS * PLANE;
    static constexpr int RED_FLOATS = CAP * NTN * 16 + CAP;
    static constexpr int LDS_FLOATS = cmax(Q_FLOATS + WP::WTH * CAP * AP, RED_FLOATS);
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
};

template <int CA, int CB, int CBS, class WP, bool GATE_P, bool GATE_Q>
__global__ __launch_bounds__(64 * WP::WTH) void k_wgrad_mfma(const float* __restrict__ Pt, const float* __restrict__ Pg,
                                                    const float* __restrict__ Qt, const float* __restrict__ Qg,
                                                    float* __restrict__ scratch, float* __restrict__ dbias_p, int B, int HP,
                                                    int HQ, int T) {
    using K = WGeo<CA, CBS, WP>;
    constexpr int NT_ = 64 * WP::WTH;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    float* xs = lds;
    float* as = lds + K::Q_FLOATS + wave * K::CAP * K::AP;
    const int b0 = blockIdx.y * CBS;
    const int tiles_h = (HP + WP::WTH - 1) / WP::WTH, tiles_t = (T + WP::WTW - 1) / WP::WTW;
    const int ntiles = B * tiles_h * tiles_t;
    const long pplane = (long)HP * T, qplane = (long)HQ * T;

    int noff[K::NTN];
#pragma unroll
    for (int nt = 0; nt < K::NTN; ++nt) {
        int n = nt * 16 + l15;
        if (n >= K::NN) n = K::NN - 1;
        const int tap = n / CBS, bl = n - tap * CBS;
        noff[nt] = bl * K::PLANE + WP::qoff(tap);
    }
    f32x4 acc[K::MT][K::NTN];
#pragma unroll
    for (int mt = 0; mt < K::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < K::NTN; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int tt = xcd_tile(tile, ntiles);
        const int tx = tt % tiles_t; tt /= tiles_t;
        const int ty = tt % tiles_h;
        const int b = tt / tiles_h, h0 = ty * WP::WTH, t0 = tx * WP::WTW;
        __syncthreads();
        {
            const long qb = ((long)b * CB + b0) * qplane;
            const int row0 = WP::q_row0(h0), col0 = t0 - WP::CH;
            for (int i = tid; i < CBS * WP::XR * WP::XC; i += NT_) {
                const int ci = i / (WP::XR * WP::XC);
                const int rem = i - ci * (WP::XR * WP::XC);
                const int r = rem / WP::XC, c = rem - r * WP::XC;
                const int h = row0 + r, t = col0 + c;
                float v = 0.f;
                if (h >= 0 && h < HQ && t >= 0 && t < T) {
                    const long o = qb + ci * qplane + (long)h * T + t;
                    v = Qt[o];
                    if (GATE_Q) v *= elu_grad_from_out(Qg[o]);
                }
                xs[ci * K::PLANE + rem] = v;
            }
        }
        {   // each wave stages its own row of P (transposed use: as[a][pixel])
            const int h = h0 + wave;
            const int t = t0 + (lane & (WP::WTW - 1));
            const bool ok = h < HP && t < T;
            const long pb = (long)b * CA * pplane + (long)(h < HP ? h : 0) * T + (t < T ? t : 0);
            // 64 lanes cover WTW pixels x (64 / WTW) channels per pass
            constexpr int CPP = 64 / WP::WTW;
            const int asub = lane / WP::WTW;
#pragma unroll 4
            for (int a0 = 0; a0 < K::CAP; a0 += CPP) {
                const int a = a0 + asub;
                float v = 0.f;
                if (ok && a < CA) {
                    const long o = pb + a * pplane;
                    v = Pt[o];
                    if (GATE_P) v *= elu_grad_from_out(Pg[o]);
                }
                as[a * K::AP + (lane & (WP::WTW - 1))] = v;
            }
        }
        __syncthreads();
        if (dbias_p && blockIdx.y == 0 && lane < CA) {
#pragma unroll 8
            for (int p = 0; p < WP::WTW; ++p) bsum += as[lane * K::AP + p];
        }
        const float* xrow = xs + WP::qrow_of_wave(wave) * WP::XC;
#pragma unroll 2
        for (int ks = 0; ks < WP::WTW / 4; ++ks) {
            float av[K::MT];
#pragma unroll
            for (int mt = 0; mt < K::MT; ++mt) av[mt] = as[(mt * 16 + l15) * K::AP + ks * 4 + g];
            const float* xp = xrow + ks * 4 + g;
#pragma unroll
            for (int nt = 0; nt < K::NTN; ++nt) {
                const float bv = xp[noff[nt]];
#pragma unroll
                for (int mt = 0; mt < K::MT; ++mt) acc[mt][nt] = mfma16(av[mt], bv, acc[mt][nt]);
            }
        }
    }
    // reduce the 4 waves of the workgroup in LDS, then write ONE partial image per workgroup to the scratch:
    // the final sum over workgroups is a second, contention-free launch (k_wgrad_reduce).
    __syncthreads();
    constexpr int NC = K::NTN * 16;
    float* red = lds;                                   // [CAP][NC] (+ CAP bias slots)
    for (int i = tid; i < K::CAP * NC + K::CAP; i += NT_) red[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < K::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < K::NTN; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(&red[(mt * 16 + 4 * g + r) * NC + nt * 16 + l15], acc[mt][nt][r]);
    if (dbias_p && blockIdx.y == 0 && lane < CA) atomicAdd(&red[K::CAP * NC + lane], bsum);
    __syncthreads();
    float* part = scratch + ((long)blockIdx.y * gridDim.x + blockIdx.x) * (K::CAP * NC);
    for (int i = tid; i < K::CAP * NC; i += NT_) part[i] = red[i];
    if (dbias_p && blockIdx.y == 0 && tid < CA) atomicAdd(dbias_p + tid, red[K::CAP * NC + tid]);
}

// LDS-DMA version (T % 4 == 0, 16-byte aligned tensors, no gating): both operands arrive by global_load_lds_dwordx4.
//   Q tile: linear [bl][row][XCP] (rows start 16-byte aligned at t0 - HL)
//   P rows: per wave [CAP][64] with the 16-byte chunks of row a stored at chunk position c ^ (a & 15) (the swizzle is
//           applied to the SOURCE address, the LDS image of a DMA is linear), so the A fragments are ds_read_b128:
//           lane (a = l15, g) owns pixels 4 sk + g, sk = 0..15 (the k order is free as long as A and B agree; consecutive
//           pixels across the four lane groups keep the B-fragment reads of the tile spread over all LDS banks).
template <int CBS, class WP>
struct WGeoD {
    static constexpr int HL = WP::CH > 0 ? 4 : 0;
    static constexpr int XCP = WP::CH > 0 ? WP::WTW + 8 : WP::WTW;
    // one 16-byte pad group per channel plane: XR * XCP is 0 or 16 mod 32, which would put the 8/16 channels that the
    // lanes of a B fragment address into one or two LDS banks; with the pad the plane pitch is 4 or 20 mod 32
    static constexpr int PLANE = WP::XR * XCP + 4;
    static constexpr int NQ = CBS * PLANE / 4;
    static constexpr int NPIECE = (NQ + 63) / 64;
    static constexpr int Q_FLOATS = NPIECE * 256;
};

// NBUF = 2: both operands double-buffered -- the next tile's DMA is in flight while this one is multiplied, one barrier per tile
// (one workgroup per CU then: the staging sets fill most of the LDS)
template <int CA, int CB, int CBS, class WP, int PREC, int NBUF>
__global__ __launch_bounds__(64 * WP::WTH) void k_wgrad_dma(const float* __restrict__ Pt, const float* __restrict__ Qt,
                                                           float* __restrict__ scratch, float* __restrict__ dbias_p, int B,
                                                           int HP, int HQ, int T) {
    using K = WGeo<CA, CBS, WP>;
    using Q = WGeoD<CBS, WP>;
    static_assert(WP::WTW == 64, "P rows are 64 pixels");
    constexpr bool BF16 = PREC != 0;
    constexpr int NT_ = 64 * WP::WTH;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    constexpr int SET = Q::Q_FLOATS + WP::WTH * K::CAP * 64;          // one staging set: Q tile + the P rows of every wave
    const int b0 = blockIdx.y * CBS;
    const int tiles_h = (HP + WP::WTH - 1) / WP::WTH, tiles_t = (T + WP::WTW - 1) / WP::WTW;
    const int ntiles = B * tiles_h * tiles_t;
    const long pplane = (long)HP * T, qplane = (long)HQ * T;

    int noff[K::NTN];
#pragma unroll
    for (int nt = 0; nt < K::NTN; ++nt) {
        int n = nt * 16 + l15;
        if (n >= K::NN) n = K::NN - 1;
        const int tap = n / CBS, bl = n - tap * CBS;
        // WP::qoff is in units of WP::XC columns per row: split it back into (row, col)
        const int qo = WP::qoff(tap), qr = qo / WP::XC, qc = qo - qr * WP::XC;
        // fp32: k-step sk of lane group g is pixel 4 sk + g (consecutive pixels across the groups: odd bank offsets)
        noff[nt] = bl * Q::PLANE + qr * Q::XCP + qc + (Q::HL - WP::CH) + (BF16 ? 8 * g : g);
    }
    f32x4 acc[K::MT][K::NTN];
#pragma unroll
    for (int mt = 0; mt < K::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < K::NTN; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    const float* zero = reinterpret_cast<const float*>(&g_zero16);

    auto issue = [&](int tile, int buf) {
        float* xs = lds + buf * SET;
        float* as = xs + Q::Q_FLOATS + uwave * K::CAP * 64;
        int tt = xcd_tile(tile, ntiles);
        const int tx = tt % tiles_t; tt /= tiles_t;
        const int ty = tt % tiles_h;
        const int b = tt / tiles_h, h0 = ty * WP::WTH, t0 = tx * WP::WTW;
        {   // Q tile
            const float* qb = Qt + ((long)b * CB + b0) * qplane;
            const int row0 = WP::q_row0(h0), col0 = t0 - Q::HL;
            constexpr int RQ = Q::XCP / 4, PQ = Q::PLANE / 4;      // 16-byte groups per row / per (padded) plane
#pragma unroll
            for (int jj = 0; jj < (Q::NPIECE + WP::WTH - 1) / WP::WTH; ++jj) {
                const int j = uwave + WP::WTH * jj;
                if (j < Q::NPIECE) {
                    const int q = j * 64 + lane;
                    const int ci = q / PQ;
                    const int rem = q - ci * PQ;
                    const int r = rem / RQ, c4 = rem - r * RQ;
                    const int h = row0 + r, t = col0 + 4 * c4;
                    const bool ok = q < Q::NQ && r < WP::XR && h >= 0 && h < HQ && t >= 0 && t < T;
                    glds16(ok ? qb + (ci * (int)qplane + h * T + t) : zero, xs + j * 256);
                }
            }
        }
        {   // this wave's row of P: 4 channels x 16 chunks per piece, chunk c of channel a lands at position c ^ (a & 15)
            const int h = h0 + wave;
            const float* pb = Pt + (long)b * CA * pplane + (long)(h < HP ? h : 0) * T + t0;
#pragma unroll
            for (int a0 = 0; a0 < K::CAP; a0 += 4) {
                const int a = a0 + g;
                const int c = l15 ^ (a & 15);
                const bool ok = a < CA && h < HP && t0 + 4 * c < T;
                glds16(ok ? pb + (a * (int)pplane + 4 * c) : zero, as + a0 * 64);
            }
        }
    };
    int buf = 0;
    if (NBUF == 2 && (int)blockIdx.x < ntiles) issue(blockIdx.x, 0);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (NBUF == 1) {
            __syncthreads();                               // everyone is done with the previous tile's LDS
            issue(tile, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // this tile has landed for every wave; the other set is free
        if (NBUF == 2 && tile + (int)gridDim.x < ntiles) issue(tile + (int)gridDim.x, buf ^ 1);
        const float* xs = lds + buf * SET;
        const float* as = xs + Q::Q_FLOATS + uwave * K::CAP * 64;
        if (dbias_p && blockIdx.y == 0 && lane < CA) {
#pragma unroll 8
            for (int p = 0; p < 64; ++p) bsum += as[lane * 64 + ((((p >> 2) ^ (lane & 15)) << 2) | (p & 3))];
        }
        const float* xrow = xs + WP::qrow_of_wave(wave) * Q::XCP;
        if constexpr (BF16) {
            // K = 32 pixels per MFMA: lane group g owns pixels 32 seg + 8 g .. + 7 (two swizzled 16-byte chunks of the P row)
#pragma unroll
            for (int seg = 0; seg < 2; ++seg) {
                bf16x8 av[K::MT], avl[K::MT];
#pragma unroll
                for (int mt = 0; mt < K::MT; ++mt) {
                    const float* row = as + (mt * 16 + l15) * 64;
                    const float4 lo = *reinterpret_cast<const float4*>(row + (((8 * seg + 2 * g) ^ l15) << 2));
                    const float4 hi = *reinterpret_cast<const float4*>(row + (((8 * seg + 2 * g + 1) ^ l15) << 2));
                    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    split_bf16<PREC>(v, av[mt], avl[mt]);
                }
#pragma unroll
                for (int nt = 0; nt < K::NTN; ++nt) {
                    const float* xp = xrow + noff[nt] + 32 * seg;
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = xp[j];
                    bf16x8 bv, bvl;
                    split_bf16<PREC>(v, bv, bvl);
#pragma unroll
                    for (int mt = 0; mt < K::MT; ++mt) acc[mt][nt] = mma_bf16<PREC>(av[mt], avl[mt], bv, bvl, acc[mt][nt]);
                }
            }
        } else {
            // A fragments: pixels 4 sk + g (sk = 0..15) of channel mt*16 + l15; 16-byte chunk sk of the row sits at sk ^ l15
            float av[K::MT][16];
#pragma unroll
            for (int mt = 0; mt < K::MT; ++mt)
#pragma unroll
                for (int sk = 0; sk < 16; ++sk) av[mt][sk] = as[(mt * 16 + l15) * 64 + ((sk ^ l15) << 2) + g];
#pragma unroll
            for (int sk = 0; sk < 16; ++sk) {
#pragma unroll
                for (int nt = 0; nt < K::NTN; ++nt) {
                    const float bv = xrow[noff[nt] + 4 * sk];
#pragma unroll
                    for (int mt = 0; mt < K::MT; ++mt) acc[mt][nt] = mfma16(av[mt][sk], bv, acc[mt][nt]);
                }
            }
        }
        if (NBUF == 2) buf ^= 1;
    }
    __syncthreads();
    constexpr int NC = K::NTN * 16;
    float* red = lds;
    for (int i = tid; i < K::CAP * NC + K::CAP; i += NT_) red[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < K::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < K::NTN; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(&red[(mt * 16 + 4 * g + r) * NC + nt * 16 + l15], acc[mt][nt][r]);
    if (dbias_p && blockIdx.y == 0 && lane < CA) atomicAdd(&red[K::CAP * NC + lane], bsum);
    __syncthreads();
    float* part = scratch + ((long)blockIdx.y * gridDim.x + blockIdx.x) * (K::CAP * NC);
    for (int i = tid; i < K::CAP * NC; i += NT_) part[i] = red[i];
    if (dbias_p && blockIdx.y == 0 && tid < CA) atomicAdd(dbias_p + tid, red[K::CAP * NC + tid]);
}

// second stage: dw[a*s_a + (y*CBS + bl)*s_b + tap*s_t] += sum over workgroups of their partial images.
// 256 threads = 32 elements x 8 partial groups: eight independent load streams per element, then an LDS reduction.
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ scratch, float* __restrict__ dw, int nblk,
                                                      int CA, int CAP, int NC, int NN, int CBS, long s_a, long s_b, long s_t) {
    __shared__ float red[8][33];
    const int e = threadIdx.x & 31, pg = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + e;
    const bool ok = i < CAP * NC;
    float s0 = 0.f, s1 = 0.f;
    if (ok) {
        const float* p = scratch + (long)blockIdx.y * nblk * (CAP * NC) + i;
        int k = pg;
        for (; k + 8 < nblk; k += 16) {
            s0 += p[(long)k * (CAP * NC)];
            s1 += p[(long)(k + 8) * (CAP * NC)];
        }
        if (k < nblk) s0 += p[(long)k * (CAP * NC)];
    }
    red[pg][e] = s0 + s1;
    __syncthreads();
    if (pg == 0 && ok) {
        const float tot = ((red[0][e] + red[1][e]) + (red[2][e] + red[3][e])) + ((red[4][e] + red[5][e]) + (red[6][e] + red[7][e]));
        const int a = i / NC, n = i - a * NC;
        if (a < CA && n < NN) {
            const int tap = n / CBS, bl = n - tap * CBS;
            dw[a * s_a + ((long)blockIdx.y * CBS + bl) * s_b + tap * s_t] += tot;
        }
    }
}

// g = dy * ELU'(y) written out AND out[c] += sum of g over (b, h, t): the gate pre-pass of the strided layers also
// produces the bias gradient, so no separate channel-sum pass is needed.  grid (C, chunks).
__global__ __launch_bounds__(256) void k_gate_and_sum(const float* __restrict__ dy, const float* __restrict__ y,
                                                      float* __restrict__ g, float* __restrict__ out, int B, int C, long inner) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    float acc = 0.f;
    const long inner4 = inner >> 2;                       // inner % 4 == 0 on this path (T % 4 == 0)
    const long total = (long)B * inner4;
    for (long i = (long)blockIdx.y * 256 + threadIdx.x; i < total; i += (long)gridDim.y * 256) {
        const long b = i / inner4, r = i - b * inner4;
        const long o = ((b * C + c) * inner >> 2) + r;
        const float4 d = reinterpret_cast<const float4*>(dy)[o], v = reinterpret_cast<const float4*>(y)[o];
        const float4 q = make_float4(d.x * elu_grad_from_out(v.x), d.y * elu_grad_from_out(v.y), d.z * elu_grad_from_out(v.z),
                                     d.w * elu_grad_from_out(v.w));
        reinterpret_cast<float4*>(g)[o] = q;
        acc += (q.x + q.y) + (q.z + q.w);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && out) atomicAdd(out + c, red[0] + red[1] + red[2] + red[3]);
}

// out[c] += sum over (b, h, t) of dy * ELU'(y)   (bias gradient of a conv + ELU layer)
__global__ __launch_bounds__(256) void k_gated_channel_sum(const float* __restrict__ dy, const float* __restrict__ y,
                                                           float* __restrict__ out, int B, int C, long inner) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    float acc = 0.f;
    const long total = (long)B * inner;
    for (long i = (long)blockIdx.y * 256 + threadIdx.x; i < total; i += (long)gridDim.y * 256) {
        const long b = i / inner, r = i - b * inner;
        const long o = (b * C + c) * inner + r;
        acc += dy[o] * elu_grad_from_out(y[o]);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out + c, red[0] + red[1] + red[2] + red[3]);
}

// ---- launch helpers ---------------------------------------------------------------------------------------------
// LDS-DMA staging needs 16-byte aligned rows (T % 4 == 0, 16-byte aligned base) and no gating while staging
inline bool dma_ok(const void* p, int T) { return (T % 4 == 0) && ((reinterpret_cast<uintptr_t>(p) & 15) == 0); }
inline int blocks_per_cu(int lds_bytes, int cap) {
    int n = (160 * 1024) / lds_bytes;
    return n > cap ? cap : (n < 1 ? 1 : n);
}
inline int persistent_grid(int ntiles, int per_cu) {
    const int cap = tt_cus() * per_cu;
    return ntiles < cap ? ntiles : cap;
}
inline int ntiles_of(int B, int H, int T) { return B * ((H + TH - 1) / TH) * ((T + TW - 1) / TW); }

template <int CIN, int COUT, class P, bool GATE, bool DMA, int PREC = 0>
int launch_conv_v(const float* x, const float* gy, const float* w, WSpec ws, const float* bias, const float* res, float* y,
                  int B, int Hin, int Hout, int T, int act, hipStream_t st) {
    using G = Geo<CIN, COUT, P, DMA, PREC>;
    constexpr int LDS = (G::W_FLOATS + G::XS_FLOATS) * 4;
    static AttrOnce attr;
    if (const int adev_ = attr.pending(); adev_ >= 0) {
        TT_HIP(hipFuncSetAttribute((const void*)k_conv_mfma<CIN, COUT, P, GATE, DMA, PREC>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr.mark(adev_);
    }
    hipLaunchKernelGGL((k_conv_mfma<CIN, COUT, P, GATE, DMA, PREC>), dim3(persistent_grid(ntiles_of(B, Hout, T), blocks_per_cu(LDS, 2))),
                       dim3(NTHREADS), LDS, st, x, gy, w, ws, bias, res, y, B, Hin, Hout, T, act);
    TT_LAUNCH_CHECK();
    return 0;
}

// ---- narrow (4,1) strided / transposed convs on the vector ALUs -----------------------------------------------------
// With at most 16 x 8 channel pairs these layers are far below the matrix ridge and a 16-row MFMA tile is mostly padding:
// like k_small_lds (conv_small.hip) the input tile (ALL input channels, P::XR rows x 64 columns) is staged by LDS-DMA,
// double-buffered, and a thread computes the COUT values of one output pixel from LDS taps with immediate offsets.
// The tap policy (which input row and weight tap output row `wave` uses) is the matrix kernels' Down4 / Up4.
template <int CIN, int COUT, class P>
struct VGeo {
    static constexpr int PLANE = P::XR * TW;
    static constexpr int NQ = CIN * PLANE / 4, NP = (NQ + 63) / 64, BUF = NP * 256;
    static constexpr int W_FLOATS = CIN * P::NWT * COUT + COUT;
    static constexpr int LDS_BYTES = (2 * BUF + W_FLOATS) * 4;
};

template <int CIN, int COUT, class P>
__global__ __launch_bounds__(NTHREADS) void k_conv_valu(const float* __restrict__ x, const float* __restrict__ w, WSpec ws,
                                                        const float* __restrict__ bias, float* __restrict__ y, int B, int Hin,
                                                        int Hout, int T, int act) {
    using V = VGeo<CIN, COUT, P>;
    static_assert(P::NTAPS != 9 && P::CH == 0, "strided geometries: no column halo");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;
    float* wimg = lds + 2 * V::BUF;                       // [c][wt][co], then the bias
    for (int i = threadIdx.x; i < CIN * P::NWT * COUT; i += NTHREADS) {
        const int co = i % COUT, wt = (i / COUT) % P::NWT, c = i / (COUT * P::NWT);
        wimg[i] = w[ws.off + co * ws.s_m + c * ws.s_c + wt * ws.s_t];
    }
    for (int i = threadIdx.x; i < COUT; i += NTHREADS) wimg[CIN * P::NWT * COUT + i] = bias ? bias[i] : 0.f;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tiles_h = (Hout + TH - 1) / TH, tiles_t = (T + TW - 1) / TW;
    const int ntiles = B * tiles_h * tiles_t;
    const long plane = (long)Hin * T, oplane = (long)Hout * T;
    const float* zero = reinterpret_cast<const float*>(&g_zero16);
    auto issue = [&](int v, int buf) {
        const Tile tl = decode_tile(v, tiles_h, tiles_t, ntiles);
        const int row0 = P::in_row0(tl.h0);
        const float* xb = x + (long)tl.b * CIN * plane;
        float* dst = xs + buf * V::BUF;
#pragma unroll
        for (int jj = 0; jj < (V::NP + 7) / 8; ++jj) {
            const int j = wave + 8 * jj;
            if (j < V::NP) {
                const int q = j * 64 + lane;
                const int ci = q / (V::PLANE / 4);
                const int rem = q - ci * (V::PLANE / 4);
                const int r = rem >> 4, c4 = rem & 15;
                const int h = row0 + r, t = tl.t0 + 4 * c4;
                const bool ok = q < V::NQ && h >= 0 && h < Hin && t < T;
                glds16(ok ? xb + (ci * (int)plane + h * T + t) : zero, dst + j * 256);
            }
        }
    };
    int v = blockIdx.x;
    __syncthreads();
    if (v >= ntiles) return;
    int buf = 0;
    issue(v, 0);
    for (; v < ntiles; v += gridDim.x) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (v + (int)gridDim.x < ntiles) issue(v + (int)gridDim.x, buf ^ 1);
        const Tile tl = decode_tile(v, tiles_h, tiles_t, ntiles);
        const float* xt = xs + buf * V::BUF + lane;
        float acc[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = wimg[CIN * P::NWT * COUT + co];
#pragma unroll 1
        for (int c = 0; c < CIN; ++c) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int tp = 0; tp < P::NTAPS; ++tp) {
                const float xv = xt[c * V::PLANE + P::lrow(tp, wave) * TW];
                const float* wl = wimg + (c * P::NWT + P::wtap(tp, wave)) * COUT;
#pragma unroll
                for (int co = 0; co < COUT; ++co) acc[co] = fmaf(xv, wl[co], acc[co]);
            }
        }
        const int h = tl.h0 + wave, t = tl.t0 + lane;
        if (h < Hout && t < T) {
            float* yb = y + (long)tl.b * COUT * oplane + (long)h * T + t;
#pragma unroll
            for (int co = 0; co < COUT; ++co) yb[co * oplane] = act == TT_ACT_ELU ? elu1(acc[co]) : acc[co];
        }
        buf ^= 1;
    }
}

template <int CIN, int COUT, class P>
int launch_conv_valu(const float* x, const float* w, WSpec ws, const float* bias, float* y, int B, int Hin, int Hout, int T, int act,
                     hipStream_t st) {
    using V = VGeo<CIN, COUT, P>;
    static AttrOnce attr;
    if (const int adev_ = attr.pending(); adev_ >= 0) {
        TT_HIP(hipFuncSetAttribute((const void*)k_conv_valu<CIN, COUT, P>, hipFuncAttributeMaxDynamicSharedMemorySize, V::LDS_BYTES));
        attr.mark(adev_);
    }
    hipLaunchKernelGGL((k_conv_valu<CIN, COUT, P>), dim3(persistent_grid(ntiles_of(B, Hout, T), blocks_per_cu(V::LDS_BYTES, 4))),
                       dim3(NTHREADS), V::LDS_BYTES, st, x, w, ws, bias, y, B, Hin, Hout, T, act);
    TT_LAUNCH_CHECK();
    return 0;
}

template <int CIN, int COUT, class P, bool GATE>
int launch_conv(const float* x, const float* gy, const float* w, WSpec ws, const float* bias, const float* res, float* y,
                int B, int Hin, int Hout, int T, int act, hipStream_t st, int prec = 0) {
    if constexpr (!GATE) {
        if (dma_ok(x, T)) {
            // measured: the transposed geometry (2 taps) wins up to 16 -> 8 channels, the strided one (4 taps) only at 4 -> 8
            if constexpr ((P::NTAPS == 2 && CIN * COUT <= 128) || (P::NTAPS == 4 && CIN * COUT <= 32)) {
                static const bool valu = !tt_tune_set("TTRAP_STRIDED_MFMA");
                if (valu && !res) return launch_conv_valu<CIN, COUT, P>(x, w, ws, bias, y, B, Hin, Hout, T, act, st);
            }
            if constexpr (P::NTAPS == 9 && CIN >= 4) {
                if (prec == 1) return launch_conv_v<CIN, COUT, P, false, true, 1>(x, gy, w, ws, bias, res, y, B, Hin, Hout, T, act, st);
                if (prec == 2) return launch_conv_v<CIN, COUT, P, false, true, 2>(x, gy, w, ws, bias, res, y, B, Hin, Hout, T, act, st);
            }
            return launch_conv_v<CIN, COUT, P, false, true>(x, gy, w, ws, bias, res, y, B, Hin, Hout, T, act, st);
        }
    }
    return launch_conv_v<CIN, COUT, P, GATE, false>(x, gy, w, ws, bias, res, y, B, Hin, Hout, T, act, st);
}

template <int C, int D, bool DMA, int PREC = 0>
int launch_rb_fwd_v(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* y, float* h1,
                    int B, int H, int T, hipStream_t st) {
    using G = Geo<C, C, Res3x3<D>, DMA, PREC>;
    using R = RB<C>;
    constexpr int LDS = (G::W_FLOATS + R::CPAD * R::CP + 2 * R::CPAD + G::XS_FLOATS) * 4;
    static AttrOnce attr;
    if (const int adev_ = attr.pending(); adev_ >= 0) {
        TT_HIP(hipFuncSetAttribute((const void*)k_rb_fwd<C, D, DMA, PREC>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr.mark(adev_);
    }
    hipLaunchKernelGGL((k_rb_fwd<C, D, DMA, PREC>), dim3(persistent_grid(ntiles_of(B, H, T), blocks_per_cu(LDS, 2))), dim3(NTHREADS), LDS,
                       st, x, w1, b1, w2, b2, y, h1, B, H, T);
    TT_LAUNCH_CHECK();
    return 0;
}

template <int C, int D>
int launch_rb_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* y, float* h1,
                  int B, int H, int T, hipStream_t st, int prec) {
    if (dma_ok(x, T)) {
        if constexpr (C >= 4) {
            if (prec == 1) return launch_rb_fwd_v<C, D, true, 1>(x, w1, b1, w2, b2, y, h1, B, H, T, st);
            if (prec == 2) return launch_rb_fwd_v<C, D, true, 2>(x, w1, b1, w2, b2, y, h1, B, H, T, st);
        }
        return launch_rb_fwd_v<C, D, true>(x, w1, b1, w2, b2, y, h1, B, H, T, st);
    }
    return launch_rb_fwd_v<C, D, false>(x, w1, b1, w2, b2, y, h1, B, H, T, st);
}

constexpr int WGRAD_MAX_BLOCKS = 512;

template <int CA, int CB, class WP, bool GP, bool GQ>
int launch_wgrad(const float* Pt, const float* Pg, const float* Qt, const float* Qg, float* dw, float* dbias_p, long s_a,
                 long s_b, long s_t, float* scratch, int B, int HP, int HQ, int T, hipStream_t st, int prec = 0) {
    constexpr int CBS = CB > 16 ? 16 : CB;
    constexpr int NS = CB / CBS;
    using K = WGeo<CA, CBS, WP>;
    const int ntiles = B * ((HP + WP::WTH - 1) / WP::WTH) * ((T + WP::WTW - 1) / WP::WTW);
    constexpr int NC = K::NTN * 16;
    int grid;
    if constexpr (!GP && !GQ) {
        if (dma_ok(Pt, T) && dma_ok(Qt, T)) {
            using Q = WGeoD<CBS, WP>;
            constexpr int SET_FLOATS = Q::Q_FLOATS + WP::WTH * K::CAP * 64;
            // Double-buffered staging (two sets, one 4-wave workgroup per CU, next tile's DMA under this tile's MFMAs) was measured
            // against the single set with two or three workgroups per CU and LOST: whole C = 32 block backward 2.20-2.31 ms against
            // 2.10-2.23 ms, C = 16 1.45-1.56 against 1.41-1.52 ms.  Several independent workgroups hide the DMA wait better than
            // one deeper pipeline.  TTRAP_WGRAD_NBUF=2 keeps the variant reachable for measurements.
            constexpr int NBUF = (CA >= 16 && 2 * SET_FLOATS * 4 <= 160 * 1024) ? 2 : 1;
            static const bool dbuf_env = NBUF == 2 && tt_tune("TTRAP_WGRAD_NBUF", 1) == 2;
            const bool dbuf = dbuf_env && !(CA >= 16 && WP::NTAPS == 9 && prec != 0);      // the bf16 modes keep the single set
            const int LDS = cmax((dbuf ? 2 : 1) * SET_FLOATS, K::RED_FLOATS) * 4;
            static AttrOnce attr;
            if (const int adev_ = attr.pending(); adev_ >= 0) {
                TT_HIP(hipFuncSetAttribute((const void*)k_wgrad_dma<CA, CB, CBS, WP, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, cmax(SET_FLOATS, K::RED_FLOATS) * 4));
                TT_HIP(hipFuncSetAttribute((const void*)k_wgrad_dma<CA, CB, CBS, WP, 0, NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize, cmax(NBUF * SET_FLOATS, K::RED_FLOATS) * 4));
                if constexpr (CA >= 16 && WP::NTAPS == 9) {
                    TT_HIP(hipFuncSetAttribute((const void*)k_wgrad_dma<CA, CB, CBS, WP, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, cmax(SET_FLOATS, K::RED_FLOATS) * 4));
                    TT_HIP(hipFuncSetAttribute((const void*)k_wgrad_dma<CA, CB, CBS, WP, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, cmax(SET_FLOATS, K::RED_FLOATS) * 4));
                }
                attr.mark(adev_);
            }
            grid = persistent_grid(ntiles, blocks_per_cu(LDS, 4));
            if (grid * NS > WGRAD_MAX_BLOCKS) grid = WGRAD_MAX_BLOCKS / NS;
            bool done = false;
            if constexpr (CA >= 16 && WP::NTAPS == 9) {
                if (prec == 1) {
                    hipLaunchKernelGGL((k_wgrad_dma<CA, CB, CBS, WP, 1, 1>), dim3(grid, NS), dim3(64 * WP::WTH), LDS, st, Pt, Qt,
                                       scratch, dbias_p, B, HP, HQ, T);
                    done = true;
                } else if (prec == 2) {
                    hipLaunchKernelGGL((k_wgrad_dma<CA, CB, CBS, WP, 2, 1>), dim3(grid, NS), dim3(64 * WP::WTH), LDS, st, Pt, Qt,
                                       scratch, dbias_p, B, HP, HQ, T);
                    done = true;
                }
            }
            if (!done) {
                if (dbuf)
                    hipLaunchKernelGGL((k_wgrad_dma<CA, CB, CBS, WP, 0, NBUF>), dim3(grid, NS), dim3(64 * WP::WTH), LDS, st, Pt, Qt,
                                       scratch, dbias_p, B, HP, HQ, T);
                else
                    hipLaunchKernelGGL((k_wgrad_dma<CA, CB, CBS, WP, 0, 1>), dim3(grid, NS), dim3(64 * WP::WTH), LDS, st, Pt, Qt,
                                       scratch, dbias_p, B, HP, HQ, T);
            }
            TT_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_wgrad_reduce, dim3((K::CAP * NC + 31) / 32, NS), dim3(256), 0, st, (const float*)scratch, dw,
                               grid, CA, K::CAP, NC, K::NN, CBS, s_a, s_b, s_t);
            TT_LAUNCH_CHECK();
            return 0;
        }
    }
    static AttrOnce attr2;
    if (const int adev_ = attr2.pending(); adev_ >= 0) {
        TT_HIP(hipFuncSetAttribute((const void*)k_wgrad_mfma<CA, CB, CBS, WP, GP, GQ>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   K::LDS_BYTES));
        attr2.mark(adev_);
    }
    grid = persistent_grid(ntiles, blocks_per_cu(K::LDS_BYTES, 4));
    if (grid * NS > WGRAD_MAX_BLOCKS) grid = WGRAD_MAX_BLOCKS / NS;
    hipLaunchKernelGGL((k_wgrad_mfma<CA, CB, CBS, WP, GP, GQ>), dim3(grid, NS), dim3(64 * WP::WTH), K::LDS_BYTES, st, Pt, Pg, Qt, Qg,
                       scratch, dbias_p, B, HP, HQ, T);
    TT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((K::CAP * NC + 31) / 32, NS), dim3(256), 0, st, (const float*)scratch, dw, grid,
                       CA, K::CAP, NC, K::NN, CBS, s_a, s_b, s_t);
    TT_LAUNCH_CHECK();
    return 0;
}

template <int C, int D, bool DMA, bool RECOMP>
int launch_rb_bwd_a_v(const float* x, const float* h1, const float* dy, const float* w1, const float* b1, const float* w2,
                      const float* b2, float* db1, float* dw2, float* db2, float* ws, int B, int H, int T, hipStream_t st) {
    using G = Geo<C, C, Res3x3<D>, DMA>;
    using R = RB<C>;
    constexpr int LDS = ((RECOMP ? G::W_FLOATS + G::XS_FLOATS : 0) + 2 * R::CPAD * R::CP + 2 * R::CPAD + R::TR_FLOATS) * 4;
    // Half-width passes (NT = 2) at 32 channels were measured and lost: even then the chain needs > 128 registers (77 spilled,
    // 312 B of scratch per lane), and the whole backward of a C = 32 block went from 2.10-2.20 ms to 2.45-2.51 ms.
    constexpr int NT = 4;
    static AttrOnce attr;
    if (const int adev_ = attr.pending(); adev_ >= 0) {
        TT_HIP(hipFuncSetAttribute((const void*)k_rb_bwd_a<C, D, DMA, RECOMP, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr.mark(adev_);
    }
    hipLaunchKernelGGL((k_rb_bwd_a<C, D, DMA, RECOMP, NT>), dim3(persistent_grid(ntiles_of(B, H, T), blocks_per_cu(LDS, 2))),
                       dim3(NTHREADS), LDS, st, x, h1, dy, w1, b1, w2, b2, ws, db1, dw2, db2, B, H, T);
    TT_LAUNCH_CHECK();
    return 0;
}

template <int C, int D>
int launch_rb_bwd(const float* x, const float* h1, const float* dy, const float* w1, const float* b1, const float* w2,
                  const float* b2, float* dx, float* dw1, float* db1, float* dw2, float* db2, float* ws, int B, int H, int T,
                  hipStream_t st, int prec) {
    int rc;
    if (h1) rc = launch_rb_bwd_a_v<C, 1, false, false>(x, h1, dy, w1, b1, w2, b2, db1, dw2, db2, ws, B, H, T, st);
    else rc = dma_ok(x, T) ? launch_rb_bwd_a_v<C, D, true, true>(x, h1, dy, w1, b1, w2, b2, db1, dw2, db2, ws, B, H, T, st)
                           : launch_rb_bwd_a_v<C, D, false, true>(x, h1, dy, w1, b1, w2, b2, db1, dw2, db2, ws, B, H, T, st);
    if (rc) return rc;
    // dx = dy + W1^T (*) dA1 : the same conv with in/out channels swapped and the taps reversed
    rc = launch_conv<C, C, Res3x3<D>, false>(ws, nullptr, w1, WSpec{9, (long)C * 9, -1, 8}, nullptr, dy, dx, B, H, H, T,
                                             TT_ACT_NONE, st, prec);
    if (rc) return rc;
    // dW1[co][ci][tap] = sum dA1[co][pix] * x[ci][pix + tap]
    return launch_wgrad<C, C, WRes<D, (C <= 8 ? 8 : 4)>, false, false>(ws, nullptr, x, nullptr, dw1, nullptr, (long)C * 9, 9, 1,
                                                                        ws + (long)B * C * H * T, B, H, H, T, st, prec);
}

// ---- 3x3 weight gradient of the narrow levels with both operands packed ----------------------------------------------
//   dW[co][ci][kh][kw] = sum_{h,t} g[co][h][t] x[ci][h + (kh-1)D][t + (kw-1)D]
// With t' = t + (kw-1)D the column shift moves from x to g:
//   A[(kw, co)][t'] = g[co][h][t' - (kw-1)D]      M = 3C rows      (three shifted copies of the wave's g row, in registers)
//   B[t'][(kh, ci)] = x[ci][h + (kh-1)D][t']      N = 3C columns   (row-shifted, column-aligned reads of the x tile)
// so C = 4 fills ONE 16x16 tile with 12x12 (k_wgrad_dma: three tiles at 4x16) and C = 8 four tiles at 24x24 (five at
// 8x16), and a k-step costs one LDS read per N tile instead of one per (tap, channel) tile.  Workgroup = 8 waves = 8 rows
// x 64 columns; x tile (row halo only) and the g rows (column halo 4) arrive by LDS-DMA; k order 4 sk + g as above.
template <int C, int D>
struct WPk {
    static constexpr int M = 3 * C, MT = (M + 15) / 16, NTN = MT;
    static constexpr int XR = 8 + 2 * D, QPLANE = XR * 64 + 4;            // padded plane pitch: 4 mod 32
    static constexpr int NQ = C * QPLANE / 4, NQP = (NQ + 63) / 64;       // x tile: 16-byte groups, wave-wide DMA pieces
    static constexpr int Q_FLOATS = NQP * 256;
    static constexpr int PROW = 72;                                      // g row: columns t0 - 4 .. t0 + 67
    static constexpr int NP = C * PROW / 4, NPP = (NP + 63) / 64;
    static constexpr int P_FLOATS = NPP * 256;                           // per wave
    static constexpr int IMG = MT * 16 * NTN * 16;                       // partial image [m][n]
    static constexpr int STAGE = Q_FLOATS + 8 * P_FLOATS;               // one staging buffer: x tile + the eight g rows
    static constexpr int LDS_FLOATS = 2 * STAGE;
    static constexpr int LDS_BYTES = (LDS_FLOATS > IMG ? LDS_FLOATS : IMG) * 4;
};

template <int C, int D>
__global__ __launch_bounds__(512) void k_wgrad3_pack(const float* __restrict__ gt, const float* __restrict__ xt,
                                                     float* __restrict__ scratch, int B, int H, int T) {
    using W = WPk<C, D>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    float* xs = lds;
    float* ps = lds + W::Q_FLOATS + wave * W::P_FLOATS;
    const int tiles_h = (H + 7) / 8, tiles_t = (T + 63) / 64;
    const int ntiles = B * tiles_h * tiles_t;
    const long plane = (long)H * T;
    const float* zero = reinterpret_cast<const float*>(&g_zero16);

    // A rows of this lane: m = mt*16 + l15 = kw*C + co ; B columns: n = nt*16 + l15 = kh*C + ci
    int aoff[W::MT], boff[W::NTN];
    bool aok[W::MT];
#pragma unroll
    for (int mt = 0; mt < W::MT; ++mt) {
        const int m = mt * 16 + l15;
        aok[mt] = m < W::M;
        const int kw = aok[mt] ? m / C : 0, co = aok[mt] ? m - kw * C : 0;
        aoff[mt] = co * W::PROW + 4 - (kw - 1) * D + g;                     // + 4 sk
        const int n = (mt * 16 + l15 < W::M) ? mt * 16 + l15 : W::M - 1;
        const int kh = n / C, ci = n - kh * C;
        boff[mt] = ci * W::QPLANE + (wave + kh * D) * 64 + g;              // + 4 sk
    }
    f32x4 acc[W::MT][W::NTN];
#pragma unroll
    for (int mt = 0; mt < W::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < W::NTN; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // both operands double-buffered: the next tile's DMA is in flight while this one is multiplied, one barrier per tile
    auto issue = [&](int tile, int buf) {
        int tt = xcd_tile(tile, ntiles);
        const int tx = tt % tiles_t; tt /= tiles_t;
        const int ty = tt % tiles_h;
        const int b = tt / tiles_h, h0 = ty * 8, t0 = tx * 64;
        float* xd = xs + buf * W::STAGE;
        float* pd = ps + buf * W::STAGE;
        {   // x tile: rows h0 - D .. h0 + 7 + D, columns t0 .. t0 + 63
            const float* xb = xt + (long)b * C * plane;
            constexpr int PQ = W::QPLANE / 4;
#pragma unroll
            for (int jj = 0; jj < (W::NQP + 7) / 8; ++jj) {
                const int j = wave + 8 * jj;
                if (j < W::NQP) {
                    const int q = j * 64 + lane;
                    const int ci = q / PQ;
                    const int rem = q - ci * PQ;
                    const int r = rem >> 4, c4 = rem & 15;
                    const int h = h0 - D + r, t = t0 + 4 * c4;
                    const bool ok = q < W::NQ && r < W::XR && h >= 0 && h < H && t < T;
                    glds16(ok ? xb + (ci * (int)plane + h * T + t) : zero, xd + j * 256);
                }
            }
        }
        {   // this wave's g row: columns t0 - 4 .. t0 + 67 of every channel
            const int h = h0 + wave;
            const float* gb = gt + (long)b * C * plane + (long)(h < H ? h : 0) * T;
#pragma unroll
            for (int jj = 0; jj < W::NPP; ++jj) {
                const int q = jj * 64 + lane;
                const int co = q / 18, c4 = q - co * 18;
                const int t = t0 - 4 + 4 * c4;
                const bool ok = q < W::NP && h < H && t >= 0 && t < T;
                glds16(ok ? gb + (co * (int)plane + t) : zero, pd + jj * 256);
            }
        }
    };
    int tile = blockIdx.x, buf = 0;
    if (tile < ntiles) issue(tile, 0);
    for (; tile < ntiles; tile += gridDim.x) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // this tile has landed, everyone is done with the other buffer
        if (tile + (int)gridDim.x < ntiles) issue(tile + (int)gridDim.x, buf ^ 1);
        const float* xsb = xs + buf * W::STAGE;
        const float* psb = ps + buf * W::STAGE;
        float av[W::MT][16];
#pragma unroll
        for (int mt = 0; mt < W::MT; ++mt)
#pragma unroll
            for (int sk = 0; sk < 16; ++sk) av[mt][sk] = aok[mt] ? psb[aoff[mt] + 4 * sk] : 0.f;
#pragma unroll
        for (int sk = 0; sk < 16; ++sk) {
#pragma unroll
            for (int nt = 0; nt < W::NTN; ++nt) {
                const float bv = xsb[boff[nt] + 4 * sk];
#pragma unroll
                for (int mt = 0; mt < W::MT; ++mt) acc[mt][nt] = mfma16(av[mt][sk], bv, acc[mt][nt]);
            }
        }
        buf ^= 1;
    }
    __syncthreads();
    float* red = lds;
    for (int i = tid; i < W::IMG; i += 512) red[i] = 0.f;
    __syncthreads();
    constexpr int NC = W::NTN * 16;
#pragma unroll
    for (int mt = 0; mt < W::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < W::NTN; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(&red[(mt * 16 + 4 * g + r) * NC + nt * 16 + l15], acc[mt][nt][r]);
    __syncthreads();
    float* part = scratch + (long)blockIdx.x * W::IMG;
    for (int i = tid; i < W::IMG; i += 512) part[i] = red[i];
}

// dw[co][ci][kh][kw] += sum over workgroups of partial[(kw*C + co)][(kh*C + ci)]
template <int C>
__global__ __launch_bounds__(256) void k_wgrad3_pack_reduce(const float* __restrict__ scratch, float* __restrict__ dw, int nblk) {
    constexpr int M = 3 * C, MT = (M + 15) / 16, NC = MT * 16, IMG = MT * 16 * NC;
    __shared__ float red[8][33];
    const int e = threadIdx.x & 31, pg = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + e;                     // element of the C*C*9 gradient
    const bool ok = i < C * C * 9;
    float s0 = 0.f;
    int co = 0, ci = 0, kh = 0, kw = 0;
    if (ok) {
        kw = i % 3; kh = (i / 3) % 3; ci = (i / 9) % C; co = i / (9 * C);
        const float* p = scratch + (kw * C + co) * NC + kh * C + ci;
        for (int k = pg; k < nblk; k += 8) s0 += p[(long)k * IMG];
    }
    red[pg][e] = s0;
    __syncthreads();
    if (pg == 0 && ok)
        dw[i] += ((red[0][e] + red[1][e]) + (red[2][e] + red[3][e])) + ((red[4][e] + red[5][e]) + (red[6][e] + red[7][e]));
}

template <int C, int D>
int launch_wgrad3_pack(const float* g, const float* x, float* dw1, float* scratch, int B, int H, int T, hipStream_t st) {
    using W = WPk<C, D>;
    static AttrOnce attr;
    if (const int adev_ = attr.pending(); adev_ >= 0) {
        TT_HIP(hipFuncSetAttribute((const void*)k_wgrad3_pack<C, D>, hipFuncAttributeMaxDynamicSharedMemorySize, W::LDS_BYTES));
        attr.mark(adev_);
    }
    const int ntiles = B * ((H + 7) / 8) * ((T + 63) / 64);
    int grid = persistent_grid(ntiles, blocks_per_cu(W::LDS_BYTES, 3));
    hipLaunchKernelGGL((k_wgrad3_pack<C, D>), dim3(grid), dim3(512), W::LDS_BYTES, st, g, x, scratch, B, H, T);
    TT_LAUNCH_CHECK();
    hipLaunchKernelGGL((k_wgrad3_pack_reduce<C>), dim3((C * C * 9 + 31) / 32), dim3(256), 0, st, (const float*)scratch, dw1, grid);
    TT_LAUNCH_CHECK();
    return 0;
}

template <int C, int D>
int rb_wgrad_only(const float* x, float* dw1, float* ws, int B, int H, int T, hipStream_t st) {
    if constexpr (C <= 8) {
        if (dma_ok(x, T) && dma_ok(ws, T)) return launch_wgrad3_pack<C, D>(ws, x, dw1, ws + (long)B * C * H * T, B, H, T, st);
    }
    return launch_wgrad<C, C, WRes<D, (C <= 8 ? 8 : 4)>, false, false>(ws, nullptr, x, nullptr, dw1, nullptr, (long)C * 9, 9, 1,
                                                                        ws + (long)B * C * H * T, B, H, H, T, st);
}

// narrow levels in the bf16 modes: pointwise chain and 3x3 data gradient on the matrix-core kernels of the wide levels, the
// weight gradient on the packed-operand kernel (fp32 MFMA: K = pixels, the operands are the fp32 tensors as they are)
template <int C, int D>
int launch_rb_bwd_narrow_mfma(const float* x, const float* h1, const float* dy, const float* w1, const float* b1, const float* w2,
                              const float* b2, float* dx, float* dw1, float* db1, float* dw2, float* db2, float* ws, int B, int H, int T,
                              hipStream_t st, int prec) {
    int rc;
    if (h1) rc = launch_rb_bwd_a_v<C, 1, false, false>(x, h1, dy, w1, b1, w2, b2, db1, dw2, db2, ws, B, H, T, st);
    else rc = dma_ok(x, T) ? launch_rb_bwd_a_v<C, D, true, true>(x, h1, dy, w1, b1, w2, b2, db1, dw2, db2, ws, B, H, T, st)
                           : launch_rb_bwd_a_v<C, D, false, true>(x, h1, dy, w1, b1, w2, b2, db1, dw2, db2, ws, B, H, T, st);
    if (rc) return rc;
    rc = launch_conv<C, C, Res3x3<D>, false>(ws, nullptr, w1, WSpec{9, (long)C * 9, -1, 8}, nullptr, dy, dx, B, H, H, T, TT_ACT_NONE, st, prec);
    if (rc) return rc;
    return rb_wgrad_only<C, D>(x, dw1, ws, B, H, T, st);
}

#define TT_DISPATCH_CD(FN, ...)                                                       \
    switch (C * 10 + dilation) {                                                      \
        case 41: return FN<4, 1>(__VA_ARGS__);   case 42: return FN<4, 2>(__VA_ARGS__);   case 43: return FN<4, 3>(__VA_ARGS__);   \
        case 81: return FN<8, 1>(__VA_ARGS__);   case 82: return FN<8, 2>(__VA_ARGS__);   case 83: return FN<8, 3>(__VA_ARGS__);   \
        case 161: return FN<16, 1>(__VA_ARGS__); case 162: return FN<16, 2>(__VA_ARGS__); case 163: return FN<16, 3>(__VA_ARGS__); \
        case 321: return FN<32, 1>(__VA_ARGS__); case 322: return FN<32, 2>(__VA_ARGS__); case 323: return FN<32, 3>(__VA_ARGS__); \
        default: return TT_E_UNSUPPORTED;                                             \
    }

extern "C" int64_t tt_wgrad_scratch_floats(void);

// the kernels index one clip (channels x rows x frames) with 32-bit element offsets
inline bool clip_fits(long channels, long H, long T) { return channels * H * T < (1L << 31); }

inline int gate_chunks(int B, long inner, int C) {
    long chunks = ((long)B * (inner >> 2) + 256L * 8 - 1) / (256L * 8);
    const long cap = 8192 / C;
    if (chunks > cap) chunks = cap;
    return chunks < 1 ? 1 : (int)chunks;
}

// strided pair: channel counts (C -> 2C)
template <int C>
int sconv_fwd(const float* x, const float* w, const float* b, float* y, int B, int H, int Hout, int T, hipStream_t st) {
    // y[a][r] = ELU(b[a] + sum_{c,kh} w[a][c][kh] x[c][2r+kh])
    return launch_conv<C, 2 * C, Down4, false>(x, nullptr, w, WSpec{(long)C * 4, 4, 1, 0}, b, nullptr, y, B, H, Hout, T, TT_ACT_ELU, st);
}
// Backward of the strided pair.  With LDS-DMA staging available the ELU' gate is applied once by a streaming
// pre-pass (g = dy * ELU'(y), into `scratch` after the reduction area) and both gradients run on the DMA kernels;
// otherwise the gate is fused into the register-staged loads.
template <int C>
int sconv_bwd(const float* x, const float* y, const float* dy, const float* w, float* dx, float* dw, float* db, float* scratch,
              int B, int H, int Hout, int T, hipStream_t st) {
    int rc = 0;
    const long ng = (long)B * 2 * C * Hout * T;
    float* g = scratch + tt_wgrad_scratch_floats();
    if (dma_ok(x, T) && dma_ok(dy, T) && dma_ok(g, T)) {
        (void)ng;
        hipLaunchKernelGGL(k_gate_and_sum, dim3(2 * C, gate_chunks(B, (long)Hout * T, 2 * C)), dim3(256), 0, st, dy, y, g, db, B,
                           2 * C, (long)Hout * T);
        TT_LAUNCH_CHECK();
        if (dx) rc = launch_conv<2 * C, C, Up4, false>(g, nullptr, w, WSpec{4, (long)C * 4, 1, 0}, nullptr, nullptr, dx, B, Hout, H, T, TT_ACT_NONE, st);
        if (rc) return rc;
        return launch_wgrad<2 * C, C, WStr<(C <= 8 ? 8 : 4)>, false, false>(g, nullptr, x, nullptr, dw, nullptr, (long)C * 4, 4, 1, scratch, B, Hout, H, T, st);
    }
    if (dx)   // dx[c][r] = sum_{a, kh: r = 2ho + kh} w[a][c][kh] * g[a][ho],  g = dy * ELU'(y)
        rc = launch_conv<2 * C, C, Up4, true>(dy, y, w, WSpec{4, (long)C * 4, 1, 0}, nullptr, nullptr, dx, B, Hout, H, T, TT_ACT_NONE, st);
    if (rc) return rc;
    // dW[a][c][kh] = sum g[a][ho] x[c][2ho+kh] ; db[a] = sum g[a]
    return launch_wgrad<2 * C, C, WStr<(C <= 8 ? 8 : 4)>, true, false>(dy, y, x, nullptr, dw, db, (long)C * 4, 4, 1, scratch, B, Hout, H, T, st);
}
template <int C>
int tconv_fwd(const float* x, const float* w, const float* b, float* y, int B, int H, int Hout, int T, hipStream_t st) {
    // y[m][r] = ELU(b[m] + sum_{a, kh: r = 2h + kh} w[a][m][kh] x[a][h])     (2C -> C)
    return launch_conv<2 * C, C, Up4, false>(x, nullptr, w, WSpec{4, (long)C * 4, 1, 0}, b, nullptr, y, B, H, Hout, T, TT_ACT_ELU, st);
}
template <int C>
int tconv_bwd(const float* x, const float* y, const float* dy, const float* w, float* dx, float* dw, float* db, float* scratch,
              int B, int H, int Hout, int T, hipStream_t st) {
    int rc = 0;
    const long ng = (long)B * C * Hout * T;
    float* g = scratch + tt_wgrad_scratch_floats();
    const bool dma = dma_ok(x, T) && dma_ok(dy, T) && dma_ok(g, T);
    if (dma) {
        (void)ng;
        hipLaunchKernelGGL(k_gate_and_sum, dim3(C, gate_chunks(B, (long)Hout * T, C)), dim3(256), 0, st, dy, y, g, db, B, C,
                           (long)Hout * T);
        TT_LAUNCH_CHECK();
        if (dx) rc = launch_conv<C, 2 * C, Down4, false>(g, nullptr, w, WSpec{(long)C * 4, 4, 1, 0}, nullptr, nullptr, dx, B, Hout, H, T, TT_ACT_NONE, st);
        if (rc) return rc;
        return launch_wgrad<2 * C, C, WStr<(C <= 8 ? 8 : 4)>, false, false>(x, nullptr, g, nullptr, dw, nullptr, (long)C * 4, 4, 1, scratch, B, H, Hout, T, st);
    }
    if (dx)   // dx[a][h] = sum_{m,kh} w[a][m][kh] g[m][2h+kh]
        rc = launch_conv<C, 2 * C, Down4, true>(dy, y, w, WSpec{(long)C * 4, 4, 1, 0}, nullptr, nullptr, dx, B, Hout, H, T, TT_ACT_NONE, st);
    if (rc) return rc;
    // dW[a][m][kh] = sum x[a][h] g[m][2h+kh]
    rc = launch_wgrad<2 * C, C, WStr<(C <= 8 ? 8 : 4)>, false, true>(x, nullptr, dy, y, dw, nullptr, (long)C * 4, 4, 1, scratch, B, H, Hout, T, st);
    if (rc) return rc;
    if (db) {
        const long inner = (long)Hout * T;
        int chunks = (int)(((long)B * inner + 256L * 16 - 1) / (256L * 16));
        const int cap = 4096 / C;
        if (chunks > cap) chunks = cap;
        if (chunks < 1) chunks = 1;
        hipLaunchKernelGGL(k_gated_channel_sum, dim3(C, chunks), dim3(256), 0, st, dy, y, db, B, C, inner);
        TT_LAUNCH_CHECK();
    }
    return 0;
}

#define TT_DISPATCH_C(FN, ...)                      \
    switch (C) {                                    \
        case 4: return FN<4>(__VA_ARGS__);          \
        case 8: return FN<8>(__VA_ARGS__);          \
        case 16: return FN<16>(__VA_ARGS__);        \
        case 32: return FN<32>(__VA_ARGS__);        \
        default: return TT_E_UNSUPPORTED;           \
    }

}  // namespace

// floats of reduction scratch the weight-gradient kernels need (per call; see include/ttrap.h)
extern "C" int64_t tt_wgrad_scratch_floats(void) { return (int64_t)WGRAD_MAX_BLOCKS * 64 * 144; }

extern "C" int tt_resblock_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                               float* y, float* h1, int B, int C, int H, int T, int dilation, int flags, void* stream) {
    if (!x || !w1 || !b1 || !w2 || !b2 || !y || B <= 0 || H <= 0 || T <= 0) return TT_E_BADARG;
    if (!clip_fits(C, H, T)) return TT_E_UNSUPPORTED;
    hipStream_t st = tt_stream(stream);
    const int bf16 = (flags & TT_FLAG_BF16_SPLIT) ? 2 : ((flags & TT_FLAG_BF16_OPERANDS) ? 1 : 0);
    // The narrow levels run on the vector ALUs in every precision mode.  Routing them through the matrix-core kernels of the
    // wide levels in the bf16 modes (padding is cheap at the bf16 rate) was measured and lost: those kernels' per-tile staging,
    // conversion and barrier costs are per PIXEL, and the narrow levels have 8-16x more pixels per channel -- C = 4: 0.68-0.79 ms
    // against 0.40 ms forward, 1.83-1.95 against 0.87-0.96 ms backward; C = 8: 0.54-0.65 against 0.55-0.62 ms forward.
    // TTRAP_NARROW_MFMA=1 keeps that route reachable for measurements.
    if (C <= 8 && (bf16 == 0 || !tt_tune_set("TTRAP_NARROW_MFMA")))
        return tt_small_rb_fwd(x, w1, b1, w2, b2, y, h1, B, C, H, T, dilation, st);
    TT_DISPATCH_CD(launch_rb_fwd, x, w1, b1, w2, b2, y, h1, B, H, T, st, bf16)
}

extern "C" int tt_resblock_bwd(const float* x, const float* h1, const float* dy, const float* w1, const float* b1,
                               const float* w2, const float* b2, float* dx, float* dw1, float* db1, float* dw2, float* db2,
                               float* ws, int B, int C, int H, int T, int dilation, int flags, void* stream) {
    if (!x || !dy || !w1 || !b1 || !w2 || !b2 || !dx || !dw1 || !db1 || !dw2 || !db2 || !ws || B <= 0 || H <= 0 || T <= 0)
        return TT_E_BADARG;
    if (!clip_fits(C, H, T)) return TT_E_UNSUPPORTED;
    hipStream_t st = tt_stream(stream);
    const int bf16n = (flags & TT_FLAG_BF16_SPLIT) ? 2 : ((flags & TT_FLAG_BF16_OPERANDS) ? 1 : 0);
    if (C <= 8 && (bf16n == 0 || !tt_tune_set("TTRAP_NARROW_MFMA"))) {
        int rc = tt_small_rb_bwd(x, h1, dy, w1, b1, w2, b2, dx, dw1, db1, dw2, db2, ws, ws + (long)B * C * H * T, B, C, H, T,
                                 dilation, st);
        if (rc == TT_SMALL_BWD_DID_DW1) return 0;           // fused narrow backward: nothing left to do
        if (rc) return rc;
        switch (C * 10 + dilation) {
            case 41: return rb_wgrad_only<4, 1>(x, dw1, ws, B, H, T, st);
            case 42: return rb_wgrad_only<4, 2>(x, dw1, ws, B, H, T, st);
            case 43: return rb_wgrad_only<4, 3>(x, dw1, ws, B, H, T, st);
            case 81: return rb_wgrad_only<8, 1>(x, dw1, ws, B, H, T, st);
            case 82: return rb_wgrad_only<8, 2>(x, dw1, ws, B, H, T, st);
            case 83: return rb_wgrad_only<8, 3>(x, dw1, ws, B, H, T, st);
            default: return TT_E_UNSUPPORTED;
        }
    }
    const int bf16 = bf16n;
    if (C <= 8) {
        switch (C * 10 + dilation) {
            case 41: return launch_rb_bwd_narrow_mfma<4, 1>(x, h1, dy, w1, b1, w2, b2, dx, dw1, db1, dw2, db2, ws, B, H, T, st, bf16);
            case 42: return launch_rb_bwd_narrow_mfma<4, 2>(x, h1, dy, w1, b1, w2, b2, dx, dw1, db1, dw2, db2, ws, B, H, T, st, bf16);
            case 43: return launch_rb_bwd_narrow_mfma<4, 3>(x, h1, dy, w1, b1, w2, b2, dx, dw1, db1, dw2, db2, ws, B, H, T, st, bf16);
            case 81: return launch_rb_bwd_narrow_mfma<8, 1>(x, h1, dy, w1, b1, w2, b2, dx, dw1, db1, dw2, db2, ws, B, H, T, st, bf16);
            case 82: return launch_rb_bwd_narrow_mfma<8, 2>(x, h1, dy, w1, b1, w2, b2, dx, dw1, db1, dw2, db2, ws, B, H, T, st, bf16);
            case 83: return launch_rb_bwd_narrow_mfma<8, 3>(x, h1, dy, w1, b1, w2, b2, dx, dw1, db1, dw2, db2, ws, B, H, T, st, bf16);
            default: return TT_E_UNSUPPORTED;
        }
    }
    TT_DISPATCH_CD(launch_rb_bwd, x, h1, dy, w1, b1, w2, b2, dx, dw1, db1, dw2, db2, ws, B, H, T, st, bf16)
}

extern "C" int tt_sconv_fwd(const float* x, const float* w, const float* b, float* y, int B, int C, int H, int T,
                            void* stream) {
    if (!x || !w || !b || !y || B <= 0 || H < 4 || T <= 0) return TT_E_BADARG;
    if (!clip_fits(2L * C, H, T)) return TT_E_UNSUPPORTED;
    const int Hout = (H - 4) / 2 + 1;
    hipStream_t st = tt_stream(stream);
    TT_DISPATCH_C(sconv_fwd, x, w, b, y, B, H, Hout, T, st)
}

extern "C" int tt_sconv_bwd(const float* x, const float* y, const float* dy, const float* w, float* dx, float* dw, float* db,
                            float* scratch, int B, int C, int H, int T, void* stream) {
    if (!x || !y || !dy || !w || !dw || !db || !scratch || B <= 0 || H < 4 || T <= 0) return TT_E_BADARG;
    if (!clip_fits(2L * C, H, T)) return TT_E_UNSUPPORTED;
    const int Hout = (H - 4) / 2 + 1;
    hipStream_t st = tt_stream(stream);
    TT_DISPATCH_C(sconv_bwd, x, y, dy, w, dx, dw, db, scratch, B, H, Hout, T, st)
}

extern "C" int tt_tconv_fwd(const float* x, const float* w, const float* b, float* y, int B, int C, int H, int T,
                            int out_pad, void* stream) {
    if (!x || !w || !b || !y || B <= 0 || H <= 0 || T <= 0 || out_pad < 0 || out_pad > 1) return TT_E_BADARG;
    if (!clip_fits(2L * C, 2L * H + 3, T)) return TT_E_UNSUPPORTED;
    const int Hout = (H - 1) * 2 + 4 + out_pad;
    hipStream_t st = tt_stream(stream);
    TT_DISPATCH_C(tconv_fwd, x, w, b, y, B, H, Hout, T, st)
}

extern "C" int tt_tconv_bwd(const float* x, const float* y, const float* dy, const float* w, float* dx, float* dw, float* db,
                            float* scratch, int B, int C, int H, int T, int out_pad, void* stream) {
    if (!x || !y || !dy || !w || !dw || !db || !scratch || B <= 0 || H <= 0 || T <= 0 || out_pad < 0 || out_pad > 1)
        return TT_E_BADARG;
    if (!clip_fits(2L * C, 2L * H + 3, T)) return TT_E_UNSUPPORTED;
    const int Hout = (H - 1) * 2 + 4 + out_pad;
    hipStream_t st = tt_stream(stream);
    TT_DISPATCH_C(tconv_bwd, x, y, dy, w, dx, dw, db, scratch, B, H, Hout, T, st)
}
