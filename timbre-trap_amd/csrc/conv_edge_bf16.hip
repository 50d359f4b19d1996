// The two 3x3 boundary convolutions of the autoencoder where fp32 planar tensors meet the bf16 channels-last interior
// (reference modules.py:433 Encoder.convin: Conv2d(2, C0, 3, 'same') + ELU;  :560 Decoder.convout: Conv2d(C0, 2, 3, 'same')),
// C0 = 4 (model_complexity 2):
//   k_cin_fwd    coefficients (B,2,H,T) fp32 planar -> ELU(conv + b) as cl16 (B,4,H,T)
//   k_cin_bwd    g = dy * ELU'(y) on the fly; dW (4,2,3,3), db; optionally dx (B,2,H,T) fp32 planar (the re-encoded transcription)
//   k_cout_fwd   cl16 (B,4,H,T) -> logits (B,2,H,T) fp32 planar = conv + b
//   k_cout_bwd   dy (B,2,H,T) fp32 planar -> dx cl16 (B,4,H,T); dW (2,4,3,3), db
// With 2 x 4 channels there is nothing for the matrix cores: a lane is a pixel, the 72 weights are wave-uniform scalars, the
// tile (16 rows x 64 frames + 1 halo) sits in LDS as fp32, and the arithmetic is fp32 throughout (only the cl16 tensors
// are bf16).  Weight / bias gradients: per-lane accumulators over the lane's pixels, reduced over the wave by shuffles and
// over the workgroup through LDS once at the end; one partial per workgroup, summed by k_edge_reduce.
#include "bf16_common.h"
#include <type_traits>

namespace {

constexpr int ETH = 16, ETW = 64, ERW = ETW + 2, EROWS = ETH + 2;      // tile and its halo
constexpr int EPLANE = EROWS * ERW;
// fp32 planar tiles in LDS: rows of PP floats = frames t0 - 4 .. t0 + 67 (sixteen-byte groups of the image row, so that a tile row is
// 18 aligned float4 loads when T % 4 == 0); frame t0 + c sits at column c + 4, the three taps of lane l at columns l + 3 .. l + 5
constexpr int PP = ETW + 8, PPLANE = EROWS * PP, PCOL0 = 3;
constexpr int NPARTW = 80;                                             // 72 weight + up to 4 bias partials, padded

__device__ __forceinline__ float gatef(float dy, float y) { return dy * elu_dout(y); }

struct ETile { int b, h0, t0; };
__device__ __forceinline__ ETile etile(int v, int tiles_h, int tiles_t, int ntiles) {
    int tile = xcd_order(v, ntiles);
    ETile r;
    r.t0 = (tile % tiles_t) * ETW; tile /= tiles_t;
    r.h0 = (tile % tiles_h) * ETH;
    r.b = tile / tiles_h;
    return r;
}

// Staging is split in two: `load` issues every global request of a tile into registers (unrolled, no branches inside), `store`
// writes them to LDS.  The kernels load tile n + 1 right after the barrier that publishes tile n and store it only after tile n's
// arithmetic, so the memory latency runs under the arithmetic instead of in front of it (10-30 registers).
// The address arithmetic was the largest single cost of these kernels (per element: two divisions by constants, two bounds tests,
// a 64-bit multiply-add chain -- about 25 four-cycle instructions for an 8-byte load): a thread's elements sit at the same place of
// every tile, so their byte offsets inside an image row block (`rel`) are computed once per kernel, and an INTERIOR tile (no halo
// element outside the image: 82 % of the tiles at the bench shape) needs one 32-bit add per element on top of a scalar base pointer.
// Tiles on the image border keep the general path.
struct TileCtx { ETile tl; bool interior, interior_v; };
__device__ __forceinline__ TileCtx tile_ctx(int v, int tiles_h, int tiles_t, int ntiles, int H, int T) {
    TileCtx c;
    c.tl = etile(v, tiles_h, tiles_t, ntiles);
    const bool rows = c.tl.h0 >= 1 && c.tl.h0 + ETH + 1 <= H;
    c.interior = rows && c.tl.t0 >= 1 && c.tl.t0 + ETW + 1 <= T;
    c.interior_v = rows && c.tl.t0 >= 4 && c.tl.t0 + ETW + 4 <= T;         // the float4 groups of a planar tile row
    return c;
}
// planes [NP][EROWS][PP] fp32 <- NP consecutive planes of a planar (B,NP,H,T) tensor, zero outside the image.
// VEC (T % 4 == 0): a thread's element is an aligned float4 of an image row -- 3 loads and 3 sixteen-byte LDS writes per thread and
// tile instead of 10 + 10 four-byte ones.
template <int NP, bool VEC>
struct StagePlanar {
    static constexpr int PER_ROW = VEC ? PP / 4 : ERW, NEL = NP * EROWS * PER_ROW, NIT = (NEL + NT - 1) / NT;
    typedef typename std::conditional<VEC, f32x4, float>::type V;
    V v[NIT];
    unsigned rel[NIT];                                           // byte offset of element `it` from the tile's first staged element
    static __device__ __forceinline__ void where(int i, int& pl, int& row, int& col) {      // col: frame - t0
        pl = i / (EROWS * PER_ROW);
        const int rem = i - pl * (EROWS * PER_ROW);
        row = rem / PER_ROW;
        col = VEC ? (rem - row * PER_ROW) * 4 - 4 : rem - row * PER_ROW - 1;
    }
    __device__ __forceinline__ void init(int H, int T, int tid) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int pl, row, col;
            where(it * NT + tid, pl, row, col);
            rel[it] = it * NT + tid < NEL ? (unsigned)((pl * H + row) * T + col + (VEC ? 4 : 1)) * 4u : 0u;
        }
    }
    __device__ __forceinline__ void load(const float* src, const TileCtx& c, int H, int T, int tid) {
        const ETile tl = c.tl;
        if (VEC ? c.interior_v : c.interior) {
            const char* base = reinterpret_cast<const char*>(src + ((long)tl.b * NP * H + (tl.h0 - 1)) * T + (tl.t0 - (VEC ? 4 : 1)));
#pragma unroll
            for (int it = 0; it < NIT; ++it) v[it] = *reinterpret_cast<const V*>(base + rel[it]);
            return;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int pl, row, col;
            where(it * NT + tid, pl, row, col);
            const int h = tl.h0 - 1 + row, t = tl.t0 + col;      // VEC: T % 4 == 0 and t % 4 == 0, so the four frames are in or out together
            const bool ok = it * NT + tid < NEL && (unsigned)h < (unsigned)H && (unsigned)t < (unsigned)T;
            const V q = *reinterpret_cast<const V*>(src + (ok ? (((long)tl.b * NP + pl) * H + h) * T + t : 0));
            if constexpr (VEC) { v[it] = ok ? q : V{0.f, 0.f, 0.f, 0.f}; } else { v[it] = ok ? q : 0.f; }
        }
    }
    __device__ __forceinline__ void store(float* lds, int tid) const {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (it * NT + tid >= NEL) continue;
            int pl, row, col;
            where(it * NT + tid, pl, row, col);
            *reinterpret_cast<V*>(lds + pl * PPLANE + row * PP + col + 4) = v[it];
        }
    }
};
// planes [4][EROWS][ERW] fp32 <- cl16 (B,H,T,4), optionally gated by the saved output
template <bool GATE>
struct StageCl4 {
    static constexpr int NIT = (EPLANE + NT - 1) / NT;
    e16x4 q[NIT], yq[GATE ? NIT : 1];
    unsigned rel[NIT];
    unsigned okm;
    __device__ __forceinline__ void init(int T, int tid) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = it * NT + tid;
            const int row = i / ERW, col = i - row * ERW;
            rel[it] = i < EPLANE ? (unsigned)(row * T + col) * 8u : 0u;
        }
    }
    __device__ __forceinline__ void load(const e16* src, const e16* ysrc, const TileCtx& c, int H, int T, int tid) {
        const ETile tl = c.tl;
        if (c.interior) {
            const long first = (((long)tl.b * H + (tl.h0 - 1)) * T + (tl.t0 - 1)) * 4;
            const char* base = reinterpret_cast<const char*>(src + first);
            const char* ybase = reinterpret_cast<const char*>(ysrc + first);
            okm = 0xffffffffu;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                q[it] = *reinterpret_cast<const e16x4*>(base + rel[it]);
                if (GATE) yq[it] = *reinterpret_cast<const e16x4*>(ybase + rel[it]);
            }
            return;
        }
        okm = 0;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = it * NT + tid;
            const int row = i / ERW, col = i - row * ERW;
            const int h = tl.h0 - 1 + row, t = tl.t0 - 1 + col;
            const bool ok = i < EPLANE && (unsigned)h < (unsigned)H && (unsigned)t < (unsigned)T;
            okm |= ok ? 1u << it : 0u;
            const long off = ok ? (((long)tl.b * H + h) * T + t) * 4 : 0;
            q[it] = *reinterpret_cast<const e16x4*>(src + off);
            if (GATE) yq[it] = *reinterpret_cast<const e16x4*>(ysrc + off);
        }
    }
    // LDS layout [pixel][4 channels] fp32: ONE 16-byte write per element here and one 16-byte read per tap pixel in the kernels (the
    // planar layout took four 4-byte writes and the staging of a tile cost as much issue time as its 144 multiply-adds:
    // convout forward 0.179 ms, 0.111 ms with the LDS writes knocked out, 0.113 ms with the multiply-adds knocked out)
    __device__ __forceinline__ void store(float* lds, int tid) const {
        const bool all = okm == 0xffffffffu;                     // uniform: an interior tile
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = it * NT + tid;
            if (i >= EPLANE) continue;
            const bool ok = all || ((okm >> it) & 1);
            f32x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float e = GATE ? gatef((float)q[it][c], (float)yq[it][c]) : (float)q[it][c];
                v[c] = ok ? e : 0.f;
            }
            *reinterpret_cast<f32x4*>(lds + i * 4) = v;
        }
    }
};

// sum `acc[N]` over the lanes of the wave, then over the four waves; thread e < N of the workgroup ends with the total
template <int N>
__device__ __forceinline__ float wg_total(float (&acc)[N], float* red, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int e = 0; e < N; ++e) {
        float s = acc[e];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) red[wave * N + e] = s;
    }
    __syncthreads();
    return tid < N ? (red[tid] + red[N + tid]) + (red[2 * N + tid] + red[3 * N + tid]) : 0.f;
}

// The multiply-adds are written on register PAIRS (v_pk_fma_f32; two output channels per instruction).  Measured on gfx950
// (tools/probes/pkfma_probe.cpp) a packed fp32 op issues at about half the rate of a plain one (5.1 vs 2.8 cycles per wave
// instruction), so the pairs are arithmetic-neutral; what they change is where the operands live.  Forward kernels pin the 36 weight
// pairs in VGPRs for the whole kernel; backward kernels (whose 72-76 gradient accumulators already fill the register file) keep a
// permuted copy of the weights in LDS and fetch one wave-uniform 8- or 16-byte group per tap, each group serving the wave's four
// rows -- as wave-uniform registers the 72 weights had been spilled to VGPR lanes and read back with one v_readlane per multiply-add.
// A wave owns ERPW CONSECUTIVE rows and carries the 3-row tap window in registers: one new row of taps per output row instead of
// three (one LDS pipe per CU serves four SIMDs).  Per call at the bench shape: convin backward 0.506 -> 0.42 ms, convout backward
// 0.399 -> 0.365 ms, the forward kernels unchanged (0.21 / 0.17 ms); 3.57 -> 3.06 ms per train step.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(float v) { f32x2 r; r[0] = v; r[1] = v; return r; }
constexpr int ERPW = ETH / 4;

// keeps the LDS reads of a row with that row (sched_barrier alone does not order loads: it touches no memory)
__device__ __forceinline__ void row_fence() { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }

// Accumulators feed only the loop-carried values, so the optimiser sinks their multiply-adds to the end of the tile body, past
// every LDS read (all windows live at once: spills).  An empty volatile asm that "modifies" them keeps each row's arithmetic in its row.
template <int N> __device__ __forceinline__ void pin(f32x2 (&a)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(a[i]));
}

// One window row = columns lane .. lane + 3 of NP planes as two register PAIRS (the fourth column is never used): a pair is what
// ds_read2_b32 returns and what v_pk_fma_f32 broadcasts from with op_sel, so no tap costs a register move.
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
template <int SEL> __device__ __forceinline__ f32x2 bcast(f32x2 v) { return __builtin_shufflevector(v, v, SEL, SEL); }
template <int NP>
struct WRow {
    f32x2 p[NP][2];
    __device__ __forceinline__ void load(const float* lds, int off) {
#pragma unroll
        for (int c = 0; c < NP; ++c) {
            p[c][0] = *reinterpret_cast<const f32x2u*>(lds + c * PPLANE + off);
            p[c][1] = *reinterpret_cast<const f32x2u*>(lds + c * PPLANE + off + 2);
        }
    }
    template <int J> __device__ __forceinline__ f32x2 tap(int c) const { return bcast<J & 1>(p[c][J >> 1]); }
    template <int J> __device__ __forceinline__ float one(int c) const { return p[c][J >> 1][J & 1]; }
};
// the rolling window: w[0..2] = halo rows r, r + 1, r + 2
template <int NP>
struct Win {
    WRow<NP> w[3];
    __device__ __forceinline__ void start(const float* lds, int r, int lane) {
        w[1].load(lds, r * PP + lane + PCOL0);
        w[2].load(lds, (r + 1) * PP + lane + PCOL0);
    }
    __device__ __forceinline__ void advance(const float* lds, int r, int lane) {      // r = the output row whose window is wanted
        w[0] = w[1];
        w[1] = w[2];
        w[2].load(lds, (r + 2) * PP + lane + PCOL0);
    }
    template <int K> __device__ __forceinline__ f32x2 tap(int c) const { return w[K / 3].template tap<K % 3>(c); }
    template <int K> __device__ __forceinline__ float one(int c) const { return w[K / 3].template one<K % 3>(c); }
};
// Four output channels per pixel = one v_mfma_f32_4x4x1_16b_f32 per (input channel, tap): sixteen independent 4 x 4 x 1 outer products
// per wave instruction, lane l in block l / 4; A = one value per lane (row l % 4), B = one value per lane (column l % 4), result
// register r of lane l = D[row r][column l % 4] (tools/probes/mfma4x4_probe.cpp).  With A = W[out channel l % 4][.][tap] (the same four
// weights in every block) and B = the lane's own pixel value, register r of lane l accumulates output channel r of pixel l: the
// thread-per-pixel loop acc[co] = fmaf(w[co], x, acc[co]) with exact fp32 products, but on the MATRIX pipe (same rate as the vector
// FMAs, csrc/conv_small.hip) -- the vector ALUs keep the weight gradient, the ELU and the conversions, and the two run side by side.
__device__ __forceinline__ f32x4 mfma441(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }
// the same for a pixel-major tile ([pixel][4] fp32): a row = three 16-byte reads, all four channels of the three tap pixels
struct PRow {
    f32x4 px[3];
    __device__ __forceinline__ void load(const float* lds, int pix) {
#pragma unroll
        for (int j = 0; j < 3; ++j) px[j] = *reinterpret_cast<const f32x4*>(lds + (pix + j) * 4);
    }
    template <int J, int C> __device__ __forceinline__ f32x2 tap() const { return __builtin_shufflevector(px[J], px[J], C, C); }
};
struct PWin {
    PRow w[3];
    __device__ __forceinline__ void start(const float* lds, int r, int lane) {
        w[1].load(lds, r * ERW + lane);
        w[2].load(lds, (r + 1) * ERW + lane);
    }
    __device__ __forceinline__ void advance(const float* lds, int r, int lane) {
        w[0] = w[1];
        w[1] = w[2];
        w[2].load(lds, (r + 2) * ERW + lane);
    }
    template <int K, int C> __device__ __forceinline__ f32x2 tap() const { return w[K / 3].template tap<K % 3, C>(); }
};
// compile-time loops: f(integral_constant)
template <int K, class F> __device__ __forceinline__ void taps9(F&& f) {
    if constexpr (K < 9) { f(std::integral_constant<int, K>{}); taps9<K + 1>(f); }
}
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// ---- Encoder.convin ------------------------------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(NT) void k_cin_fwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                 e16* __restrict__ y, int H, int T, int tiles_h, int tiles_t, int ntiles) {
    __shared__ __attribute__((aligned(16))) float xs[2 * PPLANE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float wa[18];                                                // [ci * 9 + k]: w[co = lane % 4][ci][k], the A operand of the 4x4x1 products
#pragma unroll
    for (int i = 0; i < 18; ++i) wa[i] = w[(lane & 3) * 18 + i];
    const f32x4 br = {bias[0], bias[1], bias[2], bias[3]};
    StagePlanar<2, VEC> sx;
    sx.init(H, T, tid);
    if (blockIdx.x < ntiles) sx.load(x, tile_ctx(blockIdx.x, tiles_h, tiles_t, ntiles, H, T), H, T, tid);
    for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
        const ETile tl = etile(v, tiles_h, tiles_t, ntiles);
        __syncthreads();
        sx.store(xs, tid);
        __syncthreads();
        if (v + gridDim.x < ntiles) sx.load(x, tile_ctx(v + gridDim.x, tiles_h, tiles_t, ntiles, H, T), H, T, tid);
        const int t = tl.t0 + lane, r0 = wave * ERPW;
        Win<2> xw;
        xw.start(xs, r0, lane);
        e16x4 o[ERPW];                                          // stores after the rows: no branch between them (one basic block, so
#pragma unroll                                                   // that every tap broadcast folds into the multiply-add's op_sel)
        for (int rr = 0; rr < ERPW; ++rr) {
            xw.advance(xs, r0 + rr, lane);
            f32x4 a0 = br, a1 = {0.f, 0.f, 0.f, 0.f};                 // two chains (one per input channel), all four output channels each
            taps9<0>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                a0 = mfma441(wa[k], xw.template one<k>(0), a0);
                a1 = mfma441(wa[9 + k], xw.template one<k>(1), a1);
            });
            const f32x4 a = a0 + a1;
            o[rr][0] = (e16)elu_f(a[0]); o[rr][1] = (e16)elu_f(a[1]);
            o[rr][2] = (e16)elu_f(a[2]); o[rr][3] = (e16)elu_f(a[3]);
            asm volatile("" :: "v"(o[rr]));
            row_fence();
        }
#pragma unroll
        for (int rr = 0; rr < ERPW; ++rr) {
            const int h = tl.h0 + r0 + rr;
            if (t < T && h < H) *reinterpret_cast<e16x4*>(y + (((long)tl.b * H + h) * T + t) * 4) = o[rr];
        }
    }
}

// dW[co][ci][k] = sum g[co][p] x[ci][p + k];  db[co] = sum g[co];  dx[ci][p] = sum_{co,k} W[co][ci][k] g[co][p - k]
// (g and x are zero outside the image in LDS, so only the stores are masked)
// PRE: dy arrives already gated (the first level's backward left dy * ELU'(y): tt_wide_level_bwd_gated) -- y is not read
template <bool DX, bool VEC, bool PRE = false>
__global__ __launch_bounds__(NT, 2) void k_cin_bwd(const float* __restrict__ x, const e16* __restrict__ y, const e16* __restrict__ dy,
                                                 const float* __restrict__ w, float* __restrict__ dx, float* __restrict__ part,
                                                 int H, int T, int tiles_h, int tiles_t, int ntiles, float unscale) {
    __shared__ __attribute__((aligned(16))) float xs[2 * PPLANE];
    __shared__ __attribute__((aligned(16))) float gs[4 * EPLANE + 4];
    __shared__ float red[4 * 76];
    __shared__ __attribute__((aligned(16))) float wl[72];       // [(co * 9 + k) * 2 + ci]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // dy carries the calling thread's loss scale S (a power of two): the fp32 data gradient leaves the scaled region through weights
    // multiplied by 1 / S (exact), the weight / bias sums through k_edge_reduce
    if (DX && tid < 72) { const int ci = tid & 1, q = tid >> 1; wl[tid] = unscale * w[((q / 9) * 2 + ci) * 9 + q % 9]; }
    f32x2 acc[2][18];                                            // [co pair][ci * 9 + k]: dW of co = 2p (lane 0) and 2p + 1 (lane 1)
    f32x2 accb[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int i = 0; i < 18; ++i) acc[p][i] = splat2(0.f);
        accb[p] = splat2(0.f);
    }
    StagePlanar<2, VEC> sx;
    StageCl4<!PRE> sg;
    sx.init(H, T, tid); sg.init(T, tid);
    if (blockIdx.x < ntiles) {
        const TileCtx t0 = tile_ctx(blockIdx.x, tiles_h, tiles_t, ntiles, H, T);
        sx.load(x, t0, H, T, tid); sg.load(dy, y, t0, H, T, tid);
    }
    for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
        const ETile tl = etile(v, tiles_h, tiles_t, ntiles);
        __syncthreads();
        sx.store(xs, tid);
        sg.store(gs, tid);
        __syncthreads();
        if (v + gridDim.x < ntiles) {
            const TileCtx tn = tile_ctx(v + gridDim.x, tiles_h, tiles_t, ntiles, H, T);
            sx.load(x, tn, H, T, tid); sg.load(dy, y, tn, H, T, tid);
        }
        const int t = tl.t0 + lane, r0 = wave * ERPW;
        {
            Win<2> xw;
            xw.start(xs, r0, lane);
#pragma unroll
            for (int rr = 0; rr < ERPW; ++rr) {
                const int r = r0 + rr, ctr = (r + 1) * ERW + lane + 1;
                xw.advance(xs, r, lane);
                const f32x4 gc = *reinterpret_cast<const f32x4*>(gs + ctr * 4);
                f32x2 g[2] = {__builtin_shufflevector(gc, gc, 0, 1), __builtin_shufflevector(gc, gc, 2, 3)};
                accb[0] += g[0]; accb[1] += g[1];
#pragma unroll
                for (int ci = 0; ci < 2; ++ci)
                    taps9<0>([&](auto kc) {
                        constexpr int k = decltype(kc)::value;
                        const f32x2 xv = xw.template tap<k>(ci);
                        acc[0][ci * 9 + k] = g[0] * xv + acc[0][ci * 9 + k];
                        acc[1][ci * 9 + k] = g[1] * xv + acc[1][ci * 9 + k];
                    });
                pin(acc[0]); pin(acc[1]); pin(accb);
                row_fence();
            }
        }
        if constexpr (DX) {
            f32x2 d[ERPW];                                       // (ci 0, ci 1) of the wave's rows
#pragma unroll
            for (int rr = 0; rr < ERPW; ++rr) d[rr] = splat2(0.f);
            static_for<0, 3>([&](auto jc) {                      // tap column: g at p - (k - centre) = halo row rr + 2 - k/3, column 2 - k%3
                constexpr int j = decltype(jc)::value;
                f32x4 gw[ERPW + 2];                              // all four channels of column j, halo rows r0 .. r0 + ERPW + 1
#pragma unroll
                for (int i = 0; i < ERPW + 2; ++i) gw[i] = *reinterpret_cast<const f32x4*>(gs + ((r0 + i) * ERW + lane + j) * 4);
                static_for<0, 12>([&](auto qc) {
                    constexpr int co = decltype(qc)::value / 3, kh = decltype(qc)::value % 3, k = kh * 3 + (2 - j);
                    const f32x2 w2 = *reinterpret_cast<const f32x2*>(&wl[(co * 9 + k) * 2]);
#pragma unroll
                    for (int rr = 0; rr < ERPW; ++rr)
                        d[rr] = w2 * __builtin_shufflevector(gw[rr + 2 - kh], gw[rr + 2 - kh], co, co) + d[rr];
                });
                pin(d);
                row_fence();
            });
#pragma unroll
            for (int rr = 0; rr < ERPW; ++rr) {
                const int h = tl.h0 + r0 + rr;
                if (t < T && h < H) {
                    dx[(((long)tl.b * 2 + 0) * H + h) * T + t] = d[rr][0];
                    dx[(((long)tl.b * 2 + 1) * H + h) * T + t] = d[rr][1];
                }
            }
        }
    }
    __syncthreads();
    float flat[76];                                              // back to [co][ci][k], then the 4 bias sums
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int i = 0; i < 18; ++i) { flat[(2 * p) * 18 + i] = acc[p][i][0]; flat[(2 * p + 1) * 18 + i] = acc[p][i][1]; }
        flat[72 + 2 * p] = accb[p][0]; flat[72 + 2 * p + 1] = accb[p][1];
    }
    const float tot = wg_total<76>(flat, red, tid);
    if (tid < 76) part[(long)blockIdx.x * NPARTW + tid] = tot;
}

// ---- Decoder.convout -----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_cout_fwd(const e16* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                  float* __restrict__ y, int H, int T, int tiles_h, int tiles_t, int ntiles) {
    __shared__ __attribute__((aligned(16))) float xs[4 * EPLANE + 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x2 wr[36], br;                                            // [ci * 9 + k] = (w[0][ci][k], w[1][ci][k])
#pragma unroll
    for (int i = 0; i < 36; ++i) { wr[i][0] = w[i]; wr[i][1] = w[36 + i]; asm volatile("" : "+v"(wr[i])); }
    br[0] = bias[0]; br[1] = bias[1];
    StageCl4<false> sx;
    sx.init(T, tid);
    if (blockIdx.x < ntiles) sx.load(x, nullptr, tile_ctx(blockIdx.x, tiles_h, tiles_t, ntiles, H, T), H, T, tid);
    // The results of a tile are stored one iteration LATER, just before the loads of the tile after next are issued: loads and stores
    // retire through one in-order counter, so the wait for a tile's operands also waits for every store issued before them -- stores
    // issued right after the arithmetic (a few cycles before that wait) put their whole write latency on the critical path of every tile.
    f32x2 o[ERPW];
    ETile tp; tp.b = -1; tp.h0 = tp.t0 = 0;
    const int r0 = wave * ERPW;
    auto flush = [&]() {
        if (tp.b < 0) return;
        const int t = tp.t0 + lane;
#pragma unroll
        for (int rr = 0; rr < ERPW; ++rr) {
            const int h = tp.h0 + r0 + rr;
            if (t < T && h < H) {
                y[(((long)tp.b * 2 + 0) * H + h) * T + t] = o[rr][0];
                y[(((long)tp.b * 2 + 1) * H + h) * T + t] = o[rr][1];
            }
        }
    };
    for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
        const ETile tl = etile(v, tiles_h, tiles_t, ntiles);
        __syncthreads();
        sx.store(xs, tid);
        __syncthreads();
        flush();
        if (v + gridDim.x < ntiles) sx.load(x, nullptr, tile_ctx(v + gridDim.x, tiles_h, tiles_t, ntiles, H, T), H, T, tid);
        PWin xw;
        xw.start(xs, r0, lane);
#pragma unroll
        for (int rr = 0; rr < ERPW; ++rr) {
            xw.advance(xs, r0 + rr, lane);
            f32x2 a4[4] = {br, splat2(0.f), splat2(0.f), splat2(0.f)};          // one chain per input channel
            static_for<0, 36>([&](auto ic) {
                constexpr int ci = decltype(ic)::value / 9, k = decltype(ic)::value % 9;
                a4[ci] = wr[ci * 9 + k] * xw.template tap<k, ci>() + a4[ci];
            });
            o[rr] = (a4[0] + a4[1]) + (a4[2] + a4[3]);
            asm volatile("" : "+v"(o[rr]));
            row_fence();
        }
        tp = tl;
    }
    flush();
}

template <bool VEC>
__global__ __launch_bounds__(NT, 2) void k_cout_bwd(const e16* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ w,
                                                  e16* __restrict__ dx, float* __restrict__ part, int H, int T, int tiles_h,
                                                  int tiles_t, int ntiles, float scale) {
    __shared__ __attribute__((aligned(16))) float xs[4 * EPLANE + 4];
    __shared__ __attribute__((aligned(16))) float gs[2 * PPLANE];
    __shared__ float red[4 * 74];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float wa[18];                                                // [co * 9 + k]: w[co][ci = lane % 4][k], the A operand of the data gradient
#pragma unroll
    for (int i = 0; i < 18; ++i) wa[i] = scale * w[((i / 9) * 4 + (lane & 3)) * 9 + i % 9];    // dx ENTERS the loss-scaled 16-bit region: x S (exact)
    f32x2 acc[36], accb = splat2(0.f);                           // [ci * 9 + k]: dW of co 0 (lane 0) and co 1 (lane 1)
#pragma unroll
    for (int i = 0; i < 36; ++i) acc[i] = splat2(0.f);
    StageCl4<false> sx;
    StagePlanar<2, VEC> sg;
    sx.init(T, tid); sg.init(H, T, tid);
    if (blockIdx.x < ntiles) {
        const TileCtx t0 = tile_ctx(blockIdx.x, tiles_h, tiles_t, ntiles, H, T);
        sx.load(x, nullptr, t0, H, T, tid); sg.load(dy, t0, H, T, tid);
    }
    for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
        const ETile tl = etile(v, tiles_h, tiles_t, ntiles);
        __syncthreads();
        sx.store(xs, tid);
        sg.store(gs, tid);
        __syncthreads();
        if (v + gridDim.x < ntiles) {
            const TileCtx tn = tile_ctx(v + gridDim.x, tiles_h, tiles_t, ntiles, H, T);
            sx.load(x, nullptr, tn, H, T, tid); sg.load(dy, tn, H, T, tid);
        }
        const int t = tl.t0 + lane, r0 = wave * ERPW;
        {                                                        // weight gradient on the vector ALUs: 36 packed multiply-adds per row
            PWin xw;
            xw.start(xs, r0, lane);
#pragma unroll
            for (int rr = 0; rr < ERPW; ++rr) {
                const int r = r0 + rr, ctr = (r + 1) * PP + lane + PCOL0 + 1;
                xw.advance(xs, r, lane);
                f32x2 g;
                g[0] = gs[ctr]; g[1] = gs[PPLANE + ctr];
                accb += g;
                static_for<0, 36>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    acc[i] = g * xw.template tap<i % 9, i / 9>() + acc[i];
                });
                pin(acc); asm volatile("" : "+v"(accb));
                row_fence();
            }
        }
        // data gradient dx[ci][p] = sum W[co][ci][k] dy[co][p - (k - centre)] on the matrix pipe: one 4x4x1 product per (co, tap) and row
        // gives all four input channels of the lane's pixel; the four rows are four independent accumulation chains
        f32x4 d[ERPW];
#pragma unroll
        for (int rr = 0; rr < ERPW; ++rr) d[rr] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int co = 0; co < 2; ++co) {
            WRow<1> gw[ERPW + 2];                                // halo rows r0 .. r0 + ERPW + 1 of dy[co]
#pragma unroll
            for (int i = 0; i < ERPW + 2; ++i) gw[i].load(gs + co * PPLANE, (r0 + i) * PP + lane + PCOL0);
            taps9<0>([&](auto kc) {                              // dy at p - (k - centre): halo row rr + 2 - k/3, column 2 - k%3
                constexpr int k = decltype(kc)::value;
#pragma unroll
                for (int rr = 0; rr < ERPW; ++rr) d[rr] = mfma441(wa[co * 9 + k], gw[rr + 2 - k / 3].template one<2 - k % 3>(0), d[rr]);
            });
            row_fence();
        }
#pragma unroll
        for (int rr = 0; rr < ERPW; ++rr) {
            const int h = tl.h0 + r0 + rr;
            e16x4 o;
            o[0] = (e16)d[rr][0]; o[1] = (e16)d[rr][1]; o[2] = (e16)d[rr][2]; o[3] = (e16)d[rr][3];
            if (t < T && h < H) *reinterpret_cast<e16x4*>(dx + (((long)tl.b * H + h) * T + t) * 4) = o;
        }
    }
    __syncthreads();
    float flat[74];
#pragma unroll
    for (int i = 0; i < 36; ++i) { flat[i] = acc[i][0]; flat[36 + i] = acc[i][1]; }
    flat[72] = accb[0]; flat[73] = accb[1];
    const float tot = wg_total<74>(flat, red, tid);
    if (tid < 74) part[(long)blockIdx.x * NPARTW + tid] = tot;
}

// dw[e] += sum over workgroups (e < 72), db[e - 72] likewise; 1024 threads = 64 elements x 16 slices
__global__ __launch_bounds__(1024) void k_edge_reduce(const float* __restrict__ part, int nwg, float* __restrict__ dw, float* __restrict__ db,
                                                       int nb, float scale) {
    __shared__ float red[16][64];
    const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + el;
    float s = 0.f;
    if (e < 72 + nb)
        for (int j = sl; j < nwg; j += 16) s += part[(long)j * NPARTW + e];
    red[sl][el] = s;
    __syncthreads();
    if (sl != 0 || e >= 72 + nb) return;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += red[i][el];
    if (e < 72) dw[e] += sum * scale; else db[e - 72] += sum * scale;
}

constexpr int EDGE_MAX_WG = 1024;

inline bool edge_ok(int B, int H, int T) { return B > 0 && H > 0 && T > 0 && (long)H * T * 4 < (1l << 31); }
inline void edge_tiles(int B, int H, int T, int& th, int& tt, int& n) { th = (H + ETH - 1) / ETH; tt = (T + ETW - 1) / ETW; n = B * th * tt; }
inline int edge_grid(int ntiles) { const int cap = 4 * tt_cus(); const int g = ntiles < cap ? ntiles : cap; return g < EDGE_MAX_WG ? g : EDGE_MAX_WG; }

}  // namespace

extern "C" {

int64_t tt_edge16_scratch_bytes(void) { return (int64_t)EDGE_MAX_WG * NPARTW * 4; }

int tt_convin16_fwd(const float* x, const float* w, const float* b, void* y, int B, int H, int T, void* stream) {
    if (!x || !w || !b || !y || !edge_ok(B, H, T)) return TT_E_BADARG;
    int th, tt, n; edge_tiles(B, H, T, th, tt, n);
    if (T % 4 == 0) hipLaunchKernelGGL(k_cin_fwd<true>, dim3(edge_grid(n)), dim3(NT), 0, tt_stream(stream), x, w, b, (e16*)y, H, T, th, tt, n);
    else hipLaunchKernelGGL(k_cin_fwd<false>, dim3(edge_grid(n)), dim3(NT), 0, tt_stream(stream), x, w, b, (e16*)y, H, T, th, tt, n);
    TT_LAUNCH_CHECK();
    return 0;
}

int tt_convin16_bwd(const float* x, const void* y, const void* dy, const float* w, float* dx, float* dw, float* db, void* ws, int B,
                    int H, int T, void* stream) {
    if (!x || !dy || !w || !dw || !db || !ws || !edge_ok(B, H, T)) return TT_E_BADARG;       // y == NULL: dy is already gated
    int th, tt, n; edge_tiles(B, H, T, th, tt, n);
    const int grid = edge_grid(n);
    hipStream_t st = tt_stream(stream);
    const bool vec = T % 4 == 0;
#define CIN_BWD(DX_, V_, P_) hipLaunchKernelGGL((k_cin_bwd<DX_, V_, P_>), dim3(grid), dim3(NT), 0, st, x, (const e16*)y, (const e16*)dy, w, dx, (float*)ws, H, T, th, tt, n, tt_loss_unscale())
    if (y) {
        if (dx) { if (vec) CIN_BWD(true, true, false); else CIN_BWD(true, false, false); }
        else { if (vec) CIN_BWD(false, true, false); else CIN_BWD(false, false, false); }
    } else {
        if (dx) { if (vec) CIN_BWD(true, true, true); else CIN_BWD(true, false, true); }
        else { if (vec) CIN_BWD(false, true, true); else CIN_BWD(false, false, true); }
    }
#undef CIN_BWD
    TT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_edge_reduce, dim3(2), dim3(1024), 0, st, (const float*)ws, grid, dw, db, 4, tt_loss_unscale());
    TT_LAUNCH_CHECK();
    return 0;
}

int tt_convout16_fwd(const void* x, const float* w, const float* b, float* y, int B, int H, int T, void* stream) {
    if (!x || !w || !b || !y || !edge_ok(B, H, T)) return TT_E_BADARG;
    int th, tt, n; edge_tiles(B, H, T, th, tt, n);
    hipLaunchKernelGGL(k_cout_fwd, dim3(edge_grid(n)), dim3(NT), 0, tt_stream(stream), (const e16*)x, w, b, y, H, T, th, tt, n);
    TT_LAUNCH_CHECK();
    return 0;
}

int tt_convout16_bwd(const void* x, const float* dy, const float* w, void* dx, float* dw, float* db, void* ws, int B, int H, int T,
                     void* stream) {
    if (!x || !dy || !w || !dx || !dw || !db || !ws || !edge_ok(B, H, T)) return TT_E_BADARG;
    int th, tt, n; edge_tiles(B, H, T, th, tt, n);
    const int grid = edge_grid(n);
    hipStream_t st = tt_stream(stream);
    if (T % 4 == 0) hipLaunchKernelGGL(k_cout_bwd<true>, dim3(grid), dim3(NT), 0, st, (const e16*)x, dy, w, (e16*)dx, (float*)ws, H, T, th, tt, n, tt_loss_scale());
    else hipLaunchKernelGGL(k_cout_bwd<false>, dim3(grid), dim3(NT), 0, st, (const e16*)x, dy, w, (e16*)dx, (float*)ws, H, T, th, tt, n, tt_loss_scale());
    TT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_edge_reduce, dim3(2), dim3(1024), 0, st, (const float*)ws, grid, dw, db, 2, 1.f);   // dW, db from the fp32 dy itself
    TT_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
