// Pieces shared by the 16-bit channel-innermost residual-block kernels (conv_wide_bf16.hip: one kernel per stage;
// conv_level_bf16.hip: the one-pass backward kernels -- hidden activation saved (column strips) or recomputed (tiles)) for gfx950.
#pragma once
#include "bf16_common.h"

namespace {

// Channel held by row m of co-tile ct.  C = 32: rows 4q..4q+3 of tile 0 / 1 are channels 8q..8q+3 / 8q+4..8q+7, so the
// lane that owns D rows 4g..4g+3 of both tiles owns the eight CONSECUTIVE channels 8g..8g+7 (16 bytes).  C = 16: identity.
template <int C> __device__ __forceinline__ int chan_of(int ct, int m) { return C == 32 ? 8 * (m >> 2) + 4 * ct + (m & 3) : m; }

// ---- pieces of the tile kernels that keep TWO access patterns on one LDS image (conv_level_bf16.hip; k_wrb_dxw in conv_wide_bf16.hip)
// The 16-byte channel group cg of the pixel in image column `col` sits at position cg ^ fswz(col).  C = 32: fswz = bit-reversed
// (col >> 2) & 3, so that (i) the sixteen consecutive columns a k-group reads as B operand (ds_read_b128) cover all sixteen bank
// quads and (ii) the eight consecutive pixels one half-wave addresses in a transpose read (ds_read_b64_tr_b16, 32 bytes of each)
// use alternating 32-byte halves, for ANY column alignment (the taps shift the columns by multiples of D).  C = 16: (col >> 3) & 1.
template <int C> __device__ __forceinline__ int fswz(int col) {
    return C == 32 ? ((((col >> 2) & 1) << 1) | ((col >> 3) & 1)) : ((col >> 3) & 1);
}

// Weights in MFMA operand order, bf16, written once per call by k_lvl_wprep:
//   [wf: NK x NCT x 64 lanes][wb: the same for the data gradient][w2a: NCT x 64][w2t: NCT x 64]     (16 bytes per lane)
template <int C> struct WK {
    static constexpr int NCT = C / 16;
    static constexpr int NK = C == 32 ? 9 : 5;                   // products per co-tile: one tap (C = 32) / two taps (C = 16)
    static constexpr int NCH = C == 32 ? 8 : 4;                  // channels a lane ends up with
    static constexpr int W3 = NK * NCT * 64;
    static constexpr int ENTRIES = 2 * W3 + 2 * NCT * 64;
    static constexpr int IMG_BYTES = ENTRIES * 16;
};

// 3x3 dilated product on one 16-pixel group: the lane's pixel has its top-left tap at image pixel (row, col); `img` is a
// channel-innermost LDS image IW pixels wide in the fswz layout.
template <int C, int D, int IW>
__device__ __forceinline__ void conv_taps(const unsigned char* img, int row, int col, int g,
                                          const e16x8 (&A)[WK<C>::NK][WK<C>::NCT], f32x4 (&acc)[WK<C>::NCT], e16x8& centre) {
    constexpr int NK = WK<C>::NK, NCT = WK<C>::NCT, PB = C * 2;
    const int gsel = C == 32 ? g : (g & 1);
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        int tap = C == 32 ? k : 2 * k + (g >> 1);
        if (tap > 8) tap = 8;                                    // the weights of the missing tenth tap are zero
        const int kh = tap / 3, kw = tap - 3 * kh;
        const int xc = col + kw * D;
        const e16x8 bq = *reinterpret_cast<const e16x8*>(img + ((row + kh * D) * IW + xc) * PB + 16 * (gsel ^ fswz<C>(xc)));
        if (C == 32 && k == 4) centre = bq;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ct] = mma32(A[k][ct], bq, acc[ct]);
    }
}

constexpr int MAX_A_WG = 2048, MAX_W_WG = 1024;   // workgroups that leave dumps (bounds the scratch)

// Sum of the register dumps into the fp32 gradients (+=).  1024 threads = REL consecutive dump elements x RSL slices of the
// contributing waves (bf16_common.h); the dump order keeps the loads coalesced, the scatter into dW1 / dW2 is the cheap side.
struct RedArgs {
    const float* pw; int gw;        // wgrad dumps: gw workgroups x 4 waves
    const float* pa; int ga;        // bwd_a dumps: one per workgroup
    float *dw1, *db1, *dw2, *db2;
    int split_a = 0;                // C = 32 fused backward: wave = (ci-tile, co-tile), each wave dumps only its own co-tile
    int one_dump = 0;               // the producing kernel summed its four waves: only wave slot 0 of every workgroup holds data
    float scale = tt_loss_unscale();  // 1 / S of the calling thread's loss scale (common.h): the dumps were formed from S-scaled gradients
};
// Up to four reduces in ONE launch (blockIdx.y): tt_wide_level_bwd defers the reduce of every block of a level and sums all their dumps
// at the end -- two launches (and their dependent-launch gaps) fewer per level and pass, 48 per train step.
struct RedBatch { RedArgs a[4]; int n = 0; };
template <int C>
__global__ __launch_bounds__(1024) void k_wrb_reduce(RedBatch batch) {
    const RedArgs& ar = batch.a[blockIdx.y];
    constexpr int NCT = C / 16;
    constexpr int WDUMP = 9 * NCT * 256, NEW = NCT * WDUMP;      // wgrad: elements = (wave role) x dump
    constexpr int ADUMP = C * C + 2 * C;
    __shared__ float red[RSL][REL];
    const int el = threadIdx.x % REL, sl = threadIdx.x / REL;
    const int e = blockIdx.x * REL + el;
    float sum = 0.f;
    float* dst = nullptr;
    if (e < NEW) {
        const int role = e / WDUMP, rest = e - role * WDUMP;     // C = 32: role = ci-tile = wave & 1;  C = 16: one role
        const int k = rest / (NCT * 256), a = (rest >> 8) % NCT, r = (rest >> 6) & 3, lane = rest & 63;
        const int NS = (ar.split_a || ar.one_dump) ? 1 : 4 / NCT; // waves per workgroup that contribute to this element
        const int ncontrib = ar.gw * NS;
        float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // eight loads in flight per thread
        for (int j0 = sl; j0 < ncontrib; j0 += 8 * RSL) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + RSL * u;
                if (j < ncontrib) {
                    const int wg = j / NS, s = j - wg * NS;
                    const int wave = ar.split_a ? role + 2 * a : (ar.one_dump ? 0 : (C == 32 ? role + 2 * s : s));
                    part[u] += ar.pw[((long)wg * 4 + wave) * WDUMP + rest];
                }
            }
        }
        sum = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
        const int co = 16 * a + 4 * (lane >> 4) + r, ci = 16 * role + (lane & 15);
        dst = ar.dw1 + (co * C + ci) * 9 + k;
    } else if (e < NEW + ADUMP) {
        const int q = e - NEW;
        const int ncontrib = ar.ga;
        float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int j0 = sl; j0 < ncontrib; j0 += 8 * RSL) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + RSL * u;
                if (j < ncontrib) part[u] += ar.pa[(long)j * ADUMP + q];
            }
        }
        sum = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
        if (q < C * C) {
            const int a = q / (NCT * 256), c = (q >> 8) % NCT, r = (q >> 6) & 3, lane = q & 63;
            dst = ar.dw2 + (16 * a + 4 * (lane >> 4) + r) * C + 16 * c + (lane & 15);
        } else if (q < C * C + C) dst = ar.db1 + (q - C * C);
        else dst = ar.db2 + (q - C * C - C);
    }
    red[sl][el] = sum;
    __syncthreads();
    if (sl == 0 && dst) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < RSL; ++i) s += red[i][el];
        *dst += s * ar.scale;
    }
}

// Where a launcher would start its reduce: now, or -- inside tt_wide_level_bwd -- into the level's batch (ttx_red_defer: the calling
// thread's pending batch; conv_wide_bf16.hip defines it, conv_level_bf16.hip's launchers see it too).
}  // namespace
// The backward of a weighted skip join riding on a level's gated first block (round 6; tt_wide_level_bwd_gated_join): the block's x IS the
// encoder embedding e of the join, so its GOUT epilogue -- which has x at hand for ELU'(x) -- also takes the embedding's share of the join's
// backward: dx = (dy + W1^T (*) dA1 + w * (g[0] + g[1])) * ELU'(x) with g = the gradient that reached the join's output (reps = 1 or 2 batches of
// B clips back to back, `half` elements apart), and the skip weight's gradient dw += unscale * <g[0] + g[1], x>.  Two more loads per lane and
// pixel instead of a five-tensor pass of its own (tt_skip_join16_bwd, gate & 2).  The riding kernels keep the register cap (= the occupancy) of the
// plain gated ones: left to the compiler they took 141-189 registers and a wave per SIMD less, and the skip step read 53.7-54.1 ms instead of
// 53.2-53.4 (profiles/r06_skip_ab.txt: k_wrb_bwds_sj 616 -> 457 us, k_wrb_dxw<32, SJ> 507 -> 364 us).
struct SkipJ {
    const e16* g = nullptr;        // nullptr: no join rides on this launch
    long half = 0;                 // elements between the two batches of g (0: one batch)
    const float* w = nullptr;      // &skip_weights[idx] (nullptr: 1)
    float* dw = nullptr;           // &d skip_weights[idx] (nullptr: not wanted)
    float unscale = 1.f;           // 1 / loss scale (dw leaves the 16-bit region)
};
extern thread_local SkipJ ttx_skip;         // set by tt_wide_level_bwd_gated_join; the launch that takes it resets .g
extern thread_local void* ttx_red_defer;
extern thread_local int ttx_gate_dx;        // set by tt_wide_level_bwd_gated around its first block: 1 = asked for, 2 = the block's kernel gated dx
extern thread_local bool ttx_wprep_done;     // set by tt_wide_level_bwd while the weight images of its blocks are already prepared
int ttx_wide_wprep_batch(int C, int n, const float* const* w1, const float* const* w2, void* const* ws, hipStream_t st);
namespace {
// the per-lane pieces of that epilogue: t[j] = g0[j] + g1[j]; acc += <t, x>; returns w * t[j] through `add`
template <class V, int N>
__device__ __forceinline__ void skipj_terms(const V& g0, const V& g1, bool two, const V& xq, float w, float (&add)[N], float& dot) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const float t = two ? (float)g0[j] + (float)g1[j] : (float)g0[j];
        dot = __builtin_fmaf(t, (float)xq[j], dot);
        add[j] = w * t;
    }
}
__device__ __forceinline__ void skipj_finish(float dot, const SkipJ& sj) {
    if (!sj.dw) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(sj.dw, dot * sj.unscale);
}

template <class Kernel> inline int reduce_or_defer(Kernel kern, int total, const RedArgs& ra, hipStream_t st) {
    if (ttx_red_defer) {
        RedBatch* b = static_cast<RedBatch*>(ttx_red_defer);
        if (b->n >= 4) return TT_E_BADARG;
        b->a[b->n++] = ra;
        return 0;
    }
    RedBatch one;
    one.a[0] = ra; one.n = 1;
    hipLaunchKernelGGL(kern, dim3((total + REL - 1) / REL, 1), dim3(1024), 0, st, one);
    TT_LAUNCH_CHECK();
    return 0;
}

}  // namespace
