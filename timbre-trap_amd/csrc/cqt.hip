// NSGT constant-Q transform for gfx950: forward (analysis) and inverse (synthesis).
//
// Replaces cqt_pytorch.CQT.encode/.decode as called from reference
// timbre_trap/framework/cqtwrapper.py:67 and :207, fused with to_real (:74-97), to_complex
// (:99-120) and the inf-norm of decode (:209-211).
//
// Data flow for one 66150-sample block ("block-clip" q):
//   audio viewed as Nc = 33075 complex  z[m] = x[2m] + i x[2m+1]
//   k_fft675_rows : Nc = 675 x 49 four-step FFT, step 1: 675-point FFTs as 25 x 27, both factors in registers
//                   (one LDS exchange) over n1 for 7 values of n2 per workgroup, times W_Nc^(n2 k1)
//   k_fft49_cols  : step 2: 49-point DFTs over n2 for a tile of k1 and its mirror 675-k1,
//                   then the real-FFT split -> half spectrum X[0..Nc] in natural order
//   k_band_fwd    : per (clip, bin): gather L_k windowed spectral samples straight into registers, inverse 1024-point FFT
//                   as radix 16 x 4 x 16 held in registers with two wave-local LDS transposes, one more exchange so
//                   that the re / im planes are written 16 bytes per lane (fuses CQT.to_real).  One wave per bin,
//                   4 bins per workgroup, no workgroup barrier.
// The inverse runs the same machinery backwards (k_band_inv, k_spec_gather, the same two FFT
// kernels through the conj trick, k_absmax / k_scale for the wrapper's inf-norm).
//
// HBM traffic per block-clip (algorithmic): 264,600 B audio + 4,423,680 B coefficients.  The
// spectra (2 x 264,600 B) stay in L2 / Infinity Cache between the three launches.
#include "common.h"

namespace {

constexpr int N1 = 675;          // 27 * 25
constexpr int N2 = 49;
constexpr int NC = N1 * N2;      // 33075 complex = 66150 real
constexpr int XPAD = 33088;      // padded length of one half spectrum (NC + 1 -> multiple of 32)
constexpr int M = 1024;          // frames per block (max_window_length)
constexpr int ROWS = 7;          // n2 values per workgroup in k_fft675_rows

// Step 1 of the four-step FFT.  in: [Q][NC] float2 (audio pairs, or conj(Z) for the inverse),
// out A: [Q][N2][N1] float2 = FFT_675 over n1 of in[49 n1 + n2], times W_Nc^(n2 k1).
// 675 = 25 x 27, both factors transformed in registers (one LDS exchange, one barrier pair):
//   n1 = 27 na + nb,  k1 = ka + 25 kb :   W_675^(n1 k1) = W_25^(na ka) W_675^(nb ka) W_27^(nb kb)
//   pass 1  thread (row, nb): 25-point DFT over na (5 x 5), twiddle W_675^(nb ka)      -> Ys[row][ka][nb]
//   pass 2  thread (row, ka): 27-point DFT over nb (3 x 9, 9 = 3 x 3)                  -> k1 = ka + 25 kb
// Every root of unity is read from the 675-entry table (W_5^j = tw[135 j], W_25^j = tw[27 j], W_3^j = tw[225 j],
// W_9^j = tw[75 j], W_27^j = tw[25 j]).
constexpr int FFT_ROWS_THREADS = 256;
constexpr int YS_PITCH = 28;                 // float2 per (row, ka): 27 + 1 pad

__device__ __forceinline__ void dft5_reg(float2& a0, float2& a1, float2& a2, float2& a3, float2& a4, const float2* tw) {
    const float2 w1 = tw[135], w2 = tw[270], w3 = tw[405], w4 = tw[540];
    const float2 x0 = a0, x1 = a1, x2 = a2, x3 = a3, x4 = a4;
    a0 = cadd(cadd(x0, cadd(x1, x2)), cadd(x3, x4));
    a1 = cadd(x0, cadd(cadd(cmul(x1, w1), cmul(x2, w2)), cadd(cmul(x3, w3), cmul(x4, w4))));
    a2 = cadd(x0, cadd(cadd(cmul(x1, w2), cmul(x2, w4)), cadd(cmul(x3, w1), cmul(x4, w3))));
    a3 = cadd(x0, cadd(cadd(cmul(x1, w3), cmul(x2, w1)), cadd(cmul(x3, w4), cmul(x4, w2))));
    a4 = cadd(x0, cadd(cadd(cmul(x1, w4), cmul(x2, w3)), cadd(cmul(x3, w2), cmul(x4, w1))));
}
__device__ __forceinline__ void dft3_reg(float2& a0, float2& a1, float2& a2, const float2* tw) {
    const float2 w1 = tw[225], w2 = tw[450];
    const float2 x0 = a0, x1 = a1, x2 = a2;
    a0 = cadd(x0, cadd(x1, x2));
    a1 = cadd(x0, cadd(cmul(x1, w1), cmul(x2, w2)));
    a2 = cadd(x0, cadd(cmul(x1, w2), cmul(x2, w1)));
}
// v[5 u + w] -> out[s + 5 t]   (25 = 5 x 5)
__device__ __forceinline__ void dft25_reg(float2 (&v)[25], const float2* tw) {
#pragma unroll
    for (int w = 0; w < 5; ++w) dft5_reg(v[w], v[5 + w], v[10 + w], v[15 + w], v[20 + w], tw);   // over u: result s at v[5 s + w]
#pragma unroll
    for (int s_ = 1; s_ < 5; ++s_)
#pragma unroll
        for (int w = 1; w < 5; ++w) v[5 * s_ + w] = cmul(v[5 * s_ + w], tw[27 * w * s_]);          // W_25^(w s)
    float2 o[25];
#pragma unroll
    for (int s_ = 0; s_ < 5; ++s_) {
        float2 a0 = v[5 * s_], a1 = v[5 * s_ + 1], a2 = v[5 * s_ + 2], a3 = v[5 * s_ + 3], a4 = v[5 * s_ + 4];
        dft5_reg(a0, a1, a2, a3, a4, tw);                                                            // over w: result t
        o[s_] = a0; o[s_ + 5] = a1; o[s_ + 10] = a2; o[s_ + 15] = a3; o[s_ + 20] = a4;
    }
#pragma unroll
    for (int i = 0; i < 25; ++i) v[i] = o[i];
}
// v[3 a + b] -> out[c + 3 d]   (9 = 3 x 3)
__device__ __forceinline__ void dft9_reg(float2 (&v)[9], const float2* tw) {
#pragma unroll
    for (int b = 0; b < 3; ++b) dft3_reg(v[b], v[3 + b], v[6 + b], tw);                             // over a: result c at v[3 c + b]
    v[3 * 1 + 1] = cmul(v[3 * 1 + 1], tw[75 * 1]);
    v[3 * 1 + 2] = cmul(v[3 * 1 + 2], tw[75 * 2]);
    v[3 * 2 + 1] = cmul(v[3 * 2 + 1], tw[75 * 2]);
    v[3 * 2 + 2] = cmul(v[3 * 2 + 2], tw[75 * 4]);
    float2 o[9];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float2 a0 = v[3 * c], a1 = v[3 * c + 1], a2 = v[3 * c + 2];
        dft3_reg(a0, a1, a2, tw);                                                                    // over b: result d
        o[c] = a0; o[c + 3] = a1; o[c + 6] = a2;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) v[i] = o[i];
}
// v[9 u + w] -> out[s + 3 t]   (27 = 3 x 9)
__device__ __forceinline__ void dft27_reg(float2 (&v)[27], const float2* tw) {
#pragma unroll
    for (int w = 0; w < 9; ++w) dft3_reg(v[w], v[9 + w], v[18 + w], tw);                            // over u: result s at v[9 s + w]
#pragma unroll
    for (int s_ = 1; s_ < 3; ++s_)
#pragma unroll
        for (int w = 1; w < 9; ++w) v[9 * s_ + w] = cmul(v[9 * s_ + w], tw[25 * w * s_]);           // W_27^(w s)
    float2 o[27];
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_) {
        float2 t9[9];
#pragma unroll
        for (int w = 0; w < 9; ++w) t9[w] = v[9 * s_ + w];
        dft9_reg(t9, tw);
#pragma unroll
        for (int t = 0; t < 9; ++t) o[s_ + 3 * t] = t9[t];
    }
#pragma unroll
    for (int i = 0; i < 27; ++i) v[i] = o[i];
}

__global__ __launch_bounds__(FFT_ROWS_THREADS) void k_fft675_rows(const float2* __restrict__ in, float2* __restrict__ A,
                                                                  const float2* __restrict__ tw675, const float2* __restrict__ twNc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* tw = reinterpret_cast<float2*>(smem);
    float2* Ys = tw + N1;                              // [ROWS][25][YS_PITCH]
    const int tid = threadIdx.x, g = blockIdx.x;
    const long q = blockIdx.y;
    const float2* src = in + q * NC;
    for (int i = tid; i < N1; i += FFT_ROWS_THREADS) tw[i] = tw675[i];
    __syncthreads();
    if (tid < ROWS * 27) {
        const int nb = tid / ROWS, r = tid - nb * ROWS;        // consecutive threads: consecutive n2 -> 56-byte runs
        float2 v[25];
#pragma unroll
        for (int na = 0; na < 25; ++na) v[na] = src[N2 * (27 * na + nb) + ROWS * g + r];
        dft25_reg(v, tw);
#pragma unroll
        for (int ka = 0; ka < 25; ++ka) {
            const float2 z = (ka == 0 || nb == 0) ? v[ka] : cmul(v[ka], tw[nb * ka]);       // nb ka <= 26 * 24 < 675
            Ys[(r * 25 + ka) * YS_PITCH + nb] = z;
        }
    }
    __syncthreads();
    if (tid < ROWS * 25) {
        const int r = tid / 25, ka = tid - r * 25;             // consecutive threads: consecutive k1 within a row
        float2 v[27];
        const float2* y = Ys + (r * 25 + ka) * YS_PITCH;
#pragma unroll
        for (int nb = 0; nb < 27; ++nb) v[nb] = y[nb];
        dft27_reg(v, tw);
        const int n2 = ROWS * g + r;
        float2* dst = A + q * NC + (long)n2 * N1;
#pragma unroll
        for (int kb = 0; kb < 27; ++kb) {
            const int k1 = ka + 25 * kb;
            dst[k1] = cmul(v[kb], twNc[n2 * N1 + k1]);          // table stored [n2][k1]: coalesced
        }
    }
}

// Step 2: 49-point DFTs over n2 for 16 low columns k1, their 16 mirrors 675-k1 and (tile 0) k1=0.
// 49 = 7 x 7:  n2 = 7 a + b,  k2 = e + 7 f :
//   G[b][e] = W_49^(b e) * sum_a W_7^(a e) A[7a + b]          (thread = (column, b): 7 inputs -> 7 outputs)
//   Z[e+7f] = sum_b W_7^(b f) G[b][e]                          (thread = (column, e): 7 inputs -> 7 outputs)
// MODE 0: real-FFT split -> X[q][0..NC] (half spectrum of the 66150 real samples).
// MODE 1: inverse tail  -> out[q][k] = conj(Z[k]) / NC  written as audio pairs; the workgroup's abs-max goes to *mx
//         (CQT.decode's inf-norm, reference cqtwrapper.py:209-211) with one atomic per workgroup.
constexpr int CT = 16;                      // low columns per tile
constexpr int NTILES = (337 + CT - 1) / CT; // k1 = 1..337 are "low", 338..674 their mirrors

__device__ __forceinline__ void dft7(const float2 (&in)[7], float2 (&out)[7], const float2 (&w7)[7]) {
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        float2 acc = in[0];
#pragma unroll
        for (int n = 1; n < 7; ++n) acc = cadd(acc, cmul(in[n], w7[(n * k) % 7]));
        out[k] = acc;
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_fft49_cols(const float2* __restrict__ A, float2* __restrict__ out,
                                                    const float2* __restrict__ tw49g, const float2* __restrict__ twN,
                                                    unsigned* __restrict__ mx) {
    __shared__ float2 Al[N2][2 * CT + 1];
    __shared__ float2 Zl[2 * CT + 1][N2 + 1];
    __shared__ float2 tw[N2];
    __shared__ float wmax[4];
    const int tid = threadIdx.x, tile = blockIdx.x;
    const long q = blockIdx.y;
    const float2* a = A + q * NC;
    const int ncol = (tile == 0) ? 2 * CT + 1 : 2 * CT;

    auto col_k1 = [&](int c) -> int {       // -1 = unused column
        if (c == 2 * CT) return 0;
        const int lo = tile * CT + 1 + (c & (CT - 1));
        if (lo > 337) return -1;
        return (c < CT) ? lo : N1 - lo;
    };
    if (tid < N2) tw[tid] = tw49g[tid];
    for (int i = tid; i < N2 * ncol; i += 256) {
        const int n2 = i / ncol, c = i - n2 * ncol;
        const int k1 = col_k1(c);
        Al[n2][c] = (k1 >= 0) ? a[n2 * N1 + k1] : make_float2(0.f, 0.f);
    }
    __syncthreads();
    float2 w7[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) w7[j] = tw[7 * j];
    // stage 1: thread (c, b)
    float2 gv[7];
    const bool act1 = tid < ncol * 7;
    const int c1 = tid / 7, b1 = tid - c1 * 7;
    if (act1) {
        float2 in[7];
#pragma unroll
        for (int aa = 0; aa < 7; ++aa) in[aa] = Al[7 * aa + b1][c1];
        dft7(in, gv, w7);
#pragma unroll
        for (int e = 1; e < 7; ++e) gv[e] = cmul(gv[e], tw[b1 * e]);      // W_49^(b e), b e <= 36
    }
    __syncthreads();
    if (act1) {
#pragma unroll
        for (int e = 0; e < 7; ++e) Al[7 * e + b1][c1] = gv[e];           // G[b][e] stored at row 7 e + b
    }
    __syncthreads();
    // stage 2: thread (c, e)
    if (act1) {
        const int e = b1;
        float2 in[7], o[7];
#pragma unroll
        for (int bb = 0; bb < 7; ++bb) in[bb] = Al[7 * e + bb][c1];
        dft7(in, o, w7);
#pragma unroll
        for (int f = 0; f < 7; ++f) Zl[c1][e + 7 * f] = o[f];
    }
    __syncthreads();
    if (MODE == 1) {
        const float inv = 1.0f / (float)NC;
        float2* o = out + q * NC;
        float m = 0.f;
        for (int i = tid; i < ncol * N2; i += 256) {
            const int k2 = i / ncol, c = i - k2 * ncol;
            const int k1 = col_k1(c);
            if (k1 < 0) continue;
            float2 z = Zl[c][k2];
            const float2 r = make_float2(z.x * inv, -z.y * inv);
            o[k1 + N1 * k2] = r;
            m = fmaxf(m, fmaxf(fabsf(r.x), fabsf(r.y)));
        }
        if (mx) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
            if ((tid & 63) == 0) wmax[tid >> 6] = m;
            __syncthreads();
            if (tid == 0) atomicMax(mx, __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
        }
    } else {
        float2* X = out + q * XPAD;
        // low/mirror pairs: k = k1 + 675 k2  <->  NC - k = (675 - k1) + 675 (48 - k2)
        for (int i = tid; i < CT * N2; i += 256) {
            const int k2 = i / CT, c = i - k2 * CT;
            const int k1 = col_k1(c);
            if (k1 < 0) continue;
            const int k = k1 + N1 * k2, kk = NC - k;
            const float2 za = Zl[c][k2], zb = Zl[c + CT][N2 - 1 - k2];
            {   // X[k] = (za + conj zb)/2 - i/2 W^k (za - conj zb)
                float2 e = make_float2(0.5f * (za.x + zb.x), 0.5f * (za.y - zb.y));
                float2 d = make_float2(0.5f * (za.x - zb.x), 0.5f * (za.y + zb.y));
                float2 o = cmul(make_float2(d.y, -d.x), twN[k]);     // -i * d * W^k
                X[k] = cadd(e, o);
            }
            {
                float2 e = make_float2(0.5f * (zb.x + za.x), 0.5f * (zb.y - za.y));
                float2 d = make_float2(0.5f * (zb.x - za.x), 0.5f * (zb.y + za.y));
                float2 o = cmul(make_float2(d.y, -d.x), twN[kk]);
                X[kk] = cadd(e, o);
            }
        }
        if (tile == 0) {
            for (int k2 = tid; k2 < N2; k2 += 256) {
                const float2 za = Zl[2 * CT][k2];
                if (k2 == 0) {
                    X[0] = make_float2(za.x + za.y, 0.f);
                    X[NC] = make_float2(za.x - za.y, 0.f);
                } else {
                    const float2 zb = Zl[2 * CT][N2 - k2];
                    const int k = N1 * k2;
                    float2 e = make_float2(0.5f * (za.x + zb.x), 0.5f * (za.y - zb.y));
                    float2 d = make_float2(0.5f * (za.x - zb.x), 0.5f * (za.y + zb.y));
                    float2 o = cmul(make_float2(d.y, -d.x), twN[k]);
                    X[k] = cadd(e, o);
                }
            }
        }
    }
}

// ---- 1024-point FFT, one wave per transform, data in registers -------------------------------------------------
// Lane b (0..63) holds x[64 a + b] in v[a] (a = 0..15).  1024 = 16 x 4 x 16:
//   A  per lane 16-point DFT over a            -> Y_b[d]          (d = k mod 16)
//      twiddle W_1024^(b d), transpose through LDS ([d][b], row pitch 68)
//   B  lane (q = b mod 16, dg): radix-4 over p (b = 16 p + q) for d = 4 dg + i, twiddle W_64^(q s)
//      transpose through LDS ([16 s + d][q], row pitch 18, read back with ds_read_b128)
//   C  lane j = 16 s + d: 16-point DFT over q  -> X[64 r + j] in v[r]   (natural order again)
// Only wave-local LDS hand-offs: no workgroup barrier.  INV = true conjugates every twiddle (unnormalised inverse).
constexpr int FFT_P1 = 68, FFT_P2 = 18;
constexpr int FFT_LDS_F2 = 64 * FFT_P2;          // float2 per wave (>= 16 * FFT_P1)

__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

template <bool INV>
__device__ __forceinline__ float2 twmul(float2 v, float2 w) { return INV ? cmul_conj(v, w) : cmul(v, w); }

// multiply by -i (forward) / +i (inverse)
template <bool INV>
__device__ __forceinline__ float2 mul_mi(float2 v) { return INV ? make_float2(-v.y, v.x) : make_float2(v.y, -v.x); }

template <bool INV>
__device__ __forceinline__ void dft4(float2& x0, float2& x1, float2& x2, float2& x3) {
    const float2 a = cadd(x0, x2), b = csub(x0, x2), c = cadd(x1, x3), d = mul_mi<INV>(csub(x1, x3));
    x0 = cadd(a, c); x1 = cadd(b, d); x2 = csub(a, c); x3 = csub(b, d);
}

// in-register 16-point DFT: out[k1 + 4 k2] from in[4 n1 + n2]
template <bool INV>
__device__ __forceinline__ void dft16(float2 (&v)[16]) {
    // W_16^e = (c, -s) forward
    constexpr float C1 = 0.92387953251128673848f, S1 = 0.38268343236508978178f, R2 = 0.70710678118654752440f;
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) dft4<INV>(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);     // over n1; result index k1 at v[4 k1 + n2]
    // twiddle W_16^(n2 k1)
    const float2 w1 = make_float2(C1, -S1), w2 = make_float2(R2, -R2), w3 = make_float2(S1, -C1);
    const float2 w6 = make_float2(-R2, -R2), w9 = make_float2(-C1, S1);
    v[4 * 1 + 1] = twmul<INV>(v[4 * 1 + 1], w1);
    v[4 * 1 + 2] = twmul<INV>(v[4 * 1 + 2], w2);
    v[4 * 1 + 3] = twmul<INV>(v[4 * 1 + 3], w3);
    v[4 * 2 + 1] = twmul<INV>(v[4 * 2 + 1], w2);
    v[4 * 2 + 2] = mul_mi<INV>(v[4 * 2 + 2]);                                                // W_16^4 = -i
    v[4 * 2 + 3] = twmul<INV>(v[4 * 2 + 3], w6);
    v[4 * 3 + 1] = twmul<INV>(v[4 * 3 + 1], w3);
    v[4 * 3 + 2] = twmul<INV>(v[4 * 3 + 2], w6);
    v[4 * 3 + 3] = twmul<INV>(v[4 * 3 + 3], w9);
    // over n2 for each k1: inputs v[4 k1 + n2] -> outputs k = k1 + 4 k2
    float2 o[16];
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        float2 a = v[4 * k1 + 0], b = v[4 * k1 + 1], c = v[4 * k1 + 2], d = v[4 * k1 + 3];
        dft4<INV>(a, b, c, d);
        o[k1] = a; o[k1 + 4] = b; o[k1 + 8] = c; o[k1 + 12] = d;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = o[k];
}

template <bool INV>
__device__ __forceinline__ void fft1024_wave(float2 (&v)[16], float2* lds, const float2* __restrict__ tw, int lane) {
    // ---- A
    dft16<INV>(v);
#pragma unroll
    for (int d = 1; d < 16; ++d) v[d] = twmul<INV>(v[d], tw[lane * d]);
    wave_lds_sync();
#pragma unroll
    for (int d = 0; d < 16; ++d) lds[d * FFT_P1 + lane] = v[d];
    wave_lds_sync();
    // ---- B : lane (q, dg) handles d = 4 dg + i, i = 0..3
    const int q = lane & 15, dg = lane >> 4;
    float2 u[16];                                   // u[4 i + s]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float2* row = lds + (4 * dg + i) * FFT_P1 + q;
        float2 z0 = row[0], z1 = row[16], z2 = row[32], z3 = row[48];
        dft4<INV>(z0, z1, z2, z3);
        u[4 * i + 0] = z0; u[4 * i + 1] = z1; u[4 * i + 2] = z2; u[4 * i + 3] = z3;
    }
    {
        const float2 t1 = tw[16 * q], t2 = tw[32 * q], t3 = tw[48 * q];       // W_64^(q s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u[4 * i + 1] = twmul<INV>(u[4 * i + 1], t1);
            u[4 * i + 2] = twmul<INV>(u[4 * i + 2], t2);
            u[4 * i + 3] = twmul<INV>(u[4 * i + 3], t3);
        }
    }
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int sx = 0; sx < 4; ++sx) lds[(16 * sx + 4 * dg + i) * FFT_P2 + q] = u[4 * i + sx];
    wave_lds_sync();
    // ---- C : lane j = 16 s + d reads its 16 q values (two per ds_read_b128)
    {
        const float4* row = reinterpret_cast<const float4*>(lds + lane * FFT_P2);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float4 t = row[k];
            v[2 * k] = make_float2(t.x, t.y);
            v[2 * k + 1] = make_float2(t.z, t.w);
        }
    }
    dft16<INV>(v);
}

// forward band kernel: grid (ceil(F/4), Q); one wave per bin
template <bool COMPLEX_OUT>
__global__ __launch_bounds__(256) void k_band_fwd(const float2* __restrict__ X, float* __restrict__ out,
                                                  const int4* __restrict__ bin_tab, const float* __restrict__ window,
                                                  const float2* __restrict__ tw1024, int F, int n_blocks) {
    __shared__ __attribute__((aligned(16))) float2 lds_all[4 * FFT_LDS_F2];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long q = blockIdx.y;
    const int bin = blockIdx.x * 4 + wave;
    if (bin >= F) return;                            // wave-uniform; no workgroup barrier below
    float2* lds = lds_all + wave * FFT_LDS_F2;
    const int4 bt = bin_tab[bin];                    // {spec_start, pad, length, win_off}
    const float2* x = X + q * XPAD + bt.x;
    const float* w = window + bt.w;
    float2 v[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        const int m = 64 * a + lane - bt.y;          // index inside the window
        float2 val = make_float2(0.f, 0.f);
        if (m >= 0 && m < bt.z) {
            const float2 xv = x[m];
            const float g = w[m];
            val = make_float2(xv.x * g, xv.y * g);
        }
        v[a] = val;
    }
    fft1024_wave<true>(v, lds, tw1024, lane);
    const float inv = 1.0f / (float)M;
    const long b = q / n_blocks, blk = q - b * n_blocks;
    const long Tt = (long)n_blocks * M;
    // the transform leaves frame 64 r + lane in v[r]; one more wave-local LDS exchange gives every lane four CONSECUTIVE
    // frames (256 j + 4 lane ..), so the planes are written 16 bytes per lane -- 1 KiB contiguous per wave store
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < 16; ++r) lds[64 * r + lane] = v[r];
    wave_lds_sync();
    if (COMPLEX_OUT) {
        float4* o = reinterpret_cast<float4*>(reinterpret_cast<float2*>(out) + (b * F + bin) * Tt + blk * M);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4* src = reinterpret_cast<const float4*>(lds + 256 * j + 4 * lane);
            const float4 p0 = src[0], p1 = src[1];
            o[128 * j + 2 * lane] = float4{p0.x * inv, p0.y * inv, p0.z * inv, p0.w * inv};
            o[128 * j + 2 * lane + 1] = float4{p1.x * inv, p1.y * inv, p1.z * inv, p1.w * inv};
        }
    } else {
        float4* ore = reinterpret_cast<float4*>(out + ((b * 2 + 0) * F + bin) * Tt + blk * M);
        float4* oim = reinterpret_cast<float4*>(out + ((b * 2 + 1) * F + bin) * Tt + blk * M);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4* src = reinterpret_cast<const float4*>(lds + 256 * j + 4 * lane);
            const float4 p0 = src[0], p1 = src[1];           // (re0, im0, re1, im1), (re2, im2, re3, im3)
            ore[64 * j + lane] = float4{p0.x * inv, p0.z * inv, p1.x * inv, p1.z * inv};
            oim[64 * j + lane] = float4{p0.y * inv, p0.w * inv, p1.y * inv, p1.w * inv};
        }
    }
}

// inverse band kernel: FFT of each bin's 1024 frames, times the dual window, to the ragged scratch S
template <bool COMPLEX_IN>
__global__ __launch_bounds__(256) void k_band_inv(const float* __restrict__ coeffs, float2* __restrict__ S,
                                                  const int4* __restrict__ bin_tab, const float* __restrict__ dual,
                                                  const float2* __restrict__ tw1024, int F, int n_blocks, int sum_len) {
    __shared__ __attribute__((aligned(16))) float2 lds_all[4 * FFT_LDS_F2];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long q = blockIdx.y;
    const int bin = blockIdx.x * 4 + wave;
    if (bin >= F) return;
    float2* lds = lds_all + wave * FFT_LDS_F2;
    const long b = q / n_blocks, blk = q - b * n_blocks;
    const long Tt = (long)n_blocks * M;
    float2 v[16];
    if (COMPLEX_IN) {
        const float2* c = reinterpret_cast<const float2*>(coeffs) + (b * F + bin) * Tt + blk * M;
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a] = c[64 * a + lane];
    } else {
        const float* cre = coeffs + ((b * 2 + 0) * F + bin) * Tt + blk * M;
        const float* cim = coeffs + ((b * 2 + 1) * F + bin) * Tt + blk * M;
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a] = make_float2(cre[64 * a + lane], cim[64 * a + lane]);
    }
    fft1024_wave<false>(v, lds, tw1024, lane);
    const int4 bt = bin_tab[bin];
    float2* s = S + q * sum_len + bt.w;
    const float* dl = dual + bt.w;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = 64 * r + lane - bt.y;
        if (m >= 0 && m < bt.z) {
            const float g = dl[m];
            s[m] = make_float2(v[r].x * g, v[r].y * g);
        }
    }
}

// Overlap-add of the windowed bands in the spectrum (deterministic gather through a CSR built on the
// host), then the real-IFFT pre-split:  Zc[k] = conj(E[k] + i O[k]).
__global__ __launch_bounds__(256) void k_spec_gather(const float2* __restrict__ S, float2* __restrict__ Zc,
                                                     const int* __restrict__ gat_off, const int* __restrict__ gat_idx,
                                                     const float2* __restrict__ twN, int sum_len) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= NC) return;
    const long q = blockIdx.y;
    const float2* s = S + q * sum_len;
    auto gather = [&](int j) -> float2 {
        float2 acc = make_float2(0.f, 0.f);
        for (int p = gat_off[j]; p < gat_off[j + 1]; ++p) acc = cadd(acc, s[gat_idx[p]]);
        return acc;
    };
    const float2 xa = gather(k), xb = gather(NC - k);
    const float2 e = make_float2(0.5f * (xa.x + xb.x), 0.5f * (xa.y - xb.y));
    const float2 d = make_float2(0.5f * (xa.x - xb.x), 0.5f * (xa.y + xb.y));
    const float2 o = cmul_conj(d, twN[k]);                       // d * e^{+2 pi i k / 2Nc}
    // Z = e + i*o ; store conj(Z)
    Zc[q * NC + k] = make_float2(e.x - o.y, -(e.y + o.x));
}

__global__ __launch_bounds__(256) void k_scale_by_max(float* __restrict__ x, const unsigned* __restrict__ mx, long n) {
    const float m = __uint_as_float(*mx);
    if (!(m > 0.f)) return;                                            // cqtwrapper.py:209 guard
    float2* x2 = reinterpret_cast<float2*>(x);                          // n is even (pairs of samples)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n / 2; i += (long)gridDim.x * 256) {
        float2 v = x2[i];
        x2[i] = make_float2(v.x / m, v.y / m);
    }
}

struct Scratch {
    float2* A;      // [Q][NC]
    float2* X;      // [Q][XPAD]     (forward)  /  Zc [Q][NC] (inverse)
    float2* S;      // [Q][sum_len]  (inverse)
    unsigned* mx;
};
inline int64_t align256(int64_t v) { return (v + 255) & ~int64_t(255); }
inline Scratch carve(void* base, int Q, int sum_len) {
    char* p = reinterpret_cast<char*>(base);
    Scratch s;
    s.mx = reinterpret_cast<unsigned*>(p);            p += 256;
    s.A = reinterpret_cast<float2*>(p);               p += align256((int64_t)Q * NC * 8);
    s.X = reinterpret_cast<float2*>(p);               p += align256((int64_t)Q * XPAD * 8);
    s.S = reinterpret_cast<float2*>(p);
    return s;
}

constexpr int LDS_ROWS = (N1 + ROWS * 25 * YS_PITCH) * 8;   // 44,600 B

}  // namespace

extern "C" int64_t tt_cqt_scratch_bytes(int n_clips, int n_bins, int sum_len) {
    (void)n_bins;
    return 256 + align256((int64_t)n_clips * NC * 8) + align256((int64_t)n_clips * XPAD * 8) +
           align256((int64_t)n_clips * sum_len * 8);
}

static int cqt_set_attrs() {
    static AttrOnce attr;
    if (const int adev_ = attr.pending(); adev_ >= 0) {
        TT_HIP(hipFuncSetAttribute((const void*)k_fft675_rows, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_ROWS));
        attr.mark(adev_);
    }
    return 0;
}

extern "C" int tt_cqt_forward(const tt_cqt_plan* plan, const float* audio, float* out, void* scratch,
                              int B, int n_blocks, int out_complex, void* stream) {
    if (!plan || !audio || !out || !scratch || B <= 0 || n_blocks <= 0) return TT_E_BADARG;
    int rc = cqt_set_attrs();
    if (rc) return rc;
    hipStream_t st = tt_stream(stream);
    const int Q = B * n_blocks, F = plan->n_bins;
    Scratch s = carve(scratch, Q, plan->sum_len);
    // audio (B,1,n_blocks*66150) is already [Q][66150] = [Q][NC] float2
    hipLaunchKernelGGL(k_fft675_rows, dim3(N2 / ROWS, Q), dim3(FFT_ROWS_THREADS), LDS_ROWS, st,
                       reinterpret_cast<const float2*>(audio), s.A,
                       reinterpret_cast<const float2*>(plan->tw675), reinterpret_cast<const float2*>(plan->twNc));
    TT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_fft49_cols<0>, dim3(NTILES, Q), dim3(256), 0, st, s.A, s.X,
                       reinterpret_cast<const float2*>(plan->tw49), reinterpret_cast<const float2*>(plan->twN),
                       (unsigned*)nullptr);
    TT_LAUNCH_CHECK();
    dim3 grid((F + 3) / 4, Q);
    if (out_complex)
        hipLaunchKernelGGL(k_band_fwd<true>, grid, dim3(256), 0, st, s.X, out,
                           reinterpret_cast<const int4*>(plan->bin_tab), plan->window,
                           reinterpret_cast<const float2*>(plan->tw1024), F, n_blocks);
    else
        hipLaunchKernelGGL(k_band_fwd<false>, grid, dim3(256), 0, st, s.X, out,
                           reinterpret_cast<const int4*>(plan->bin_tab), plan->window,
                           reinterpret_cast<const float2*>(plan->tw1024), F, n_blocks);
    TT_LAUNCH_CHECK();
    return 0;
}

extern "C" int tt_cqt_inverse(const tt_cqt_plan* plan, const float* coeffs, float* audio, void* scratch,
                              int B, int n_blocks, int in_complex, int normalize, void* stream) {
    if (!plan || !coeffs || !audio || !scratch || B <= 0 || n_blocks <= 0) return TT_E_BADARG;
    int rc = cqt_set_attrs();
    if (rc) return rc;
    hipStream_t st = tt_stream(stream);
    const int Q = B * n_blocks, F = plan->n_bins;
    Scratch s = carve(scratch, Q, plan->sum_len);
    dim3 grid((F + 3) / 4, Q);
    if (in_complex)
        hipLaunchKernelGGL(k_band_inv<true>, grid, dim3(256), 0, st, coeffs, s.S,
                           reinterpret_cast<const int4*>(plan->bin_tab), plan->dual,
                           reinterpret_cast<const float2*>(plan->tw1024), F, n_blocks, plan->sum_len);
    else
        hipLaunchKernelGGL(k_band_inv<false>, grid, dim3(256), 0, st, coeffs, s.S,
                           reinterpret_cast<const int4*>(plan->bin_tab), plan->dual,
                           reinterpret_cast<const float2*>(plan->tw1024), F, n_blocks, plan->sum_len);
    TT_LAUNCH_CHECK();
    float2* Zc = s.X;   // reuse: [Q][NC]
    hipLaunchKernelGGL(k_spec_gather, dim3((NC + 255) / 256, Q), dim3(256), 0, st, s.S, Zc, plan->gat_off,
                       plan->gat_idx, reinterpret_cast<const float2*>(plan->twN), plan->sum_len);
    TT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_fft675_rows, dim3(N2 / ROWS, Q), dim3(FFT_ROWS_THREADS), LDS_ROWS, st, Zc, s.A,
                       reinterpret_cast<const float2*>(plan->tw675), reinterpret_cast<const float2*>(plan->twNc));
    TT_LAUNCH_CHECK();
    if (normalize) TT_HIP(hipMemsetAsync(s.mx, 0, 4, st));
    hipLaunchKernelGGL(k_fft49_cols<1>, dim3(NTILES, Q), dim3(256), 0, st, s.A, reinterpret_cast<float2*>(audio),
                       reinterpret_cast<const float2*>(plan->tw49), reinterpret_cast<const float2*>(plan->twN),
                       normalize ? s.mx : (unsigned*)nullptr);
    TT_LAUNCH_CHECK();
    if (normalize) {
        const long n = (long)Q * 2 * NC;
        hipLaunchKernelGGL(k_scale_by_max, dim3(2048), dim3(256), 0, st, audio, s.mx, n);
        TT_LAUNCH_CHECK();
    }
    return 0;
}
