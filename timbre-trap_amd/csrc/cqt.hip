// NSGT constant-Q transform for gfx950: forward (analysis) and inverse (synthesis).
//
// Replaces cqt_pytorch.CQT.encode/.decode as called from reference
// timbre_trap/framework/cqtwrapper.py:67 and :207, fused with to_real (:74-97), to_complex
// (:99-120) and the inf-norm of decode (:209-211).
//
// Data flow for one 66150-sample block ("block-clip" q):
//   audio viewed as Nc = 33075 complex  z[m] = x[2m] + i x[2m+1]
//   k_fft675_rows : Nc = 675 x 49 four-step FFT, step 1: 675-point Stockham FFTs (radix 5,5,3,3,3)
//                   in LDS over n1 for 7 values of n2 per workgroup, times W_Nc^(n2 k1)
//   k_fft49_cols  : step 2: 49-point DFTs over n2 for a tile of k1 and its mirror 675-k1,
//                   then the real-FFT split -> half spectrum X[0..Nc] in natural order
//   k_band_fwd    : per (clip, bin): gather L_k windowed spectral samples into a zeroed 1024
//                   buffer, radix-4 Stockham inverse FFT in LDS, write re / im planes (fuses
//                   CQT.to_real).  One wave per bin, 4 bins per workgroup.
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

template <int R> struct Roots;
template <> struct Roots<3> {
    static __device__ __forceinline__ float2 w(int i) {
        const float c[3] = {1.f, -0.5f, -0.5f};
        const float s[3] = {0.f, -0.86602540378443864676f, 0.86602540378443864676f};
        return make_float2(c[i], s[i]);
    }
};
template <> struct Roots<5> {
    static __device__ __forceinline__ float2 w(int i) {
        const float c[5] = {1.f, 0.30901699437494742410f, -0.80901699437494742410f,
                            -0.80901699437494742410f, 0.30901699437494742410f};
        const float s[5] = {0.f, -0.95105651629515357212f, -0.58778525229247312917f,
                            0.58778525229247312917f, 0.95105651629515357212f};
        return make_float2(c[i], s[i]);
    }
};

// forward DFT of R points, direct
template <int R>
__device__ __forceinline__ void dft_small(float2 (&v)[R]) {
    float2 o[R];
#pragma unroll
    for (int a = 0; a < R; ++a) {
        float2 acc = v[0];
#pragma unroll
        for (int b = 1; b < R; ++b) acc = cadd(acc, cmul(v[b], Roots<R>::w((a * b) % R)));
        o[a] = acc;
    }
#pragma unroll
    for (int a = 0; a < R; ++a) v[a] = o[a];
}

// One Stockham stage of radix R over ROWS independent 675-point rows held in LDS.
template <int R>
__device__ __forceinline__ void stage675(const float2* __restrict__ in, float2* __restrict__ out, int Ns,
                                         const float2* __restrict__ tw, int tid, int nthreads) {
    constexpr int NB = N1 / R;                 // butterflies per row
    const int step = N1 / (Ns * R);
    for (int wi = tid; wi < ROWS * NB; wi += nthreads) {
        const int row = wi / NB, j = wi - row * NB;
        const int k = j % Ns;
        float2 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float2 x = in[row * N1 + j + r * NB];
            v[r] = (r == 0) ? x : cmul(x, tw[r * k * step]);
        }
        dft_small<R>(v);
        const int j0 = (j / Ns) * Ns * R + k;
#pragma unroll
        for (int r = 0; r < R; ++r) out[row * N1 + j0 + r * Ns] = v[r];
    }
    __syncthreads();
}

// Step 1 of the four-step FFT.  in: [Q][NC] float2 (audio pairs, or conj(Z) for the inverse),
// out A: [Q][N2][N1] float2 = FFT_675 over n1 of in[49 n1 + n2], times W_Nc^(n2 k1).
__global__ __launch_bounds__(256) void k_fft675_rows(const float2* __restrict__ in, float2* __restrict__ A,
                                                     const float2* __restrict__ tw675, const float2* __restrict__ twNc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* buf0 = reinterpret_cast<float2*>(smem);
    float2* buf1 = buf0 + ROWS * N1;
    float2* tw = buf1 + ROWS * N1;
    const int tid = threadIdx.x, g = blockIdx.x;
    const long q = blockIdx.y;
    const float2* src = in + q * NC;
    for (int i = tid; i < N1; i += 256) tw[i] = tw675[i];
    for (int i = tid; i < ROWS * N1; i += 256) {
        const int n1 = i / ROWS, r = i - n1 * ROWS;
        buf0[r * N1 + n1] = src[N2 * n1 + ROWS * g + r];
    }
    __syncthreads();
    stage675<5>(buf0, buf1, 1, tw, tid, 256);
    stage675<5>(buf1, buf0, 5, tw, tid, 256);
    stage675<3>(buf0, buf1, 25, tw, tid, 256);
    stage675<3>(buf1, buf0, 75, tw, tid, 256);
    stage675<3>(buf0, buf1, 225, tw, tid, 256);
    float2* dst = A + q * NC;
    for (int i = tid; i < ROWS * N1; i += 256) {
        const int r = i / N1, k1 = i - r * N1;
        const int n2 = ROWS * g + r;
        dst[n2 * N1 + k1] = cmul(buf1[i], twNc[n2 * k1]);
    }
}

// Step 2: 49-point DFTs over n2 for 16 low columns k1, their 16 mirrors 675-k1 and (tile 0) k1=0.
// MODE 0: real-FFT split -> X[q][0..NC] (half spectrum of the 66150 real samples).
// MODE 1: inverse tail  -> out[q][k] = conj(Z[k]) / NC  written as audio pairs.
constexpr int CT = 16;                      // low columns per tile
constexpr int NTILES = (337 + CT - 1) / CT; // k1 = 1..337 are "low", 338..674 their mirrors
template <int MODE>
__global__ __launch_bounds__(256) void k_fft49_cols(const float2* __restrict__ A, float2* __restrict__ out,
                                                    const float2* __restrict__ tw49g, const float2* __restrict__ twN) {
    __shared__ float2 Al[N2][2 * CT + 1];
    __shared__ float2 Zl[2 * CT + 1][N2 + 1];
    __shared__ float2 tw[N2];
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
    for (int i = tid; i < ncol * N2; i += 256) {
        const int c = i / N2, k2 = i - c * N2;
        float2 acc = make_float2(0.f, 0.f);
        int e = 0;                          // (n2 * k2) mod 49
#pragma unroll 7
        for (int n2 = 0; n2 < N2; ++n2) {
            acc = cadd(acc, cmul(Al[n2][c], tw[e]));
            e += k2; if (e >= N2) e -= N2;
        }
        Zl[c][k2] = acc;
    }
    __syncthreads();
    if (MODE == 1) {
        const float inv = 1.0f / (float)NC;
        float2* o = out + q * NC;
        for (int i = tid; i < ncol * N2; i += 256) {
            const int k2 = i / ncol, c = i - k2 * ncol;
            const int k1 = col_k1(c);
            if (k1 < 0) continue;
            float2 z = Zl[c][k2];
            o[k1 + N1 * k2] = make_float2(z.x * inv, -z.y * inv);
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

// ---- 1024-point radix-4 Stockham FFT, one wave per transform, ping-pong in LDS ----------------
// INV = true uses conjugate twiddles (unnormalised inverse).
template <bool INV>
__device__ __forceinline__ void fft1024_stage(const float2* __restrict__ in, float2* __restrict__ out, int Ns,
                                              const float2* __restrict__ tw, int lane) {
    const int step = M / (Ns * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = lane + 64 * i;
        const int k = j & (Ns - 1);
        float2 v0 = in[j], v1 = in[j + 256], v2 = in[j + 512], v3 = in[j + 768];
        if (Ns > 1) {
            const float2 w1 = tw[k * step], w2 = tw[2 * k * step], w3 = tw[3 * k * step];
            if (INV) { v1 = cmul_conj(v1, w1); v2 = cmul_conj(v2, w2); v3 = cmul_conj(v3, w3); }
            else     { v1 = cmul(v1, w1);      v2 = cmul(v2, w2);      v3 = cmul(v3, w3); }
        }
        const float2 a = cadd(v0, v2), b = csub(v0, v2), c = cadd(v1, v3), d = csub(v1, v3);
        // forward: -i*d = (d.y, -d.x); inverse: +i*d = (-d.y, d.x)
        const float2 id = INV ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x);
        const int j0 = ((j - k) << 2) + k;
        out[j0] = cadd(a, c);
        out[j0 + Ns] = cadd(b, id);
        out[j0 + 2 * Ns] = csub(a, c);
        out[j0 + 3 * Ns] = csub(b, id);
    }
}

template <bool INV>
__device__ __forceinline__ float2* fft1024(float2* buf0, float2* buf1, const float2* tw, int lane) {
    __syncthreads();
    fft1024_stage<INV>(buf0, buf1, 1, tw, lane);   __syncthreads();
    fft1024_stage<INV>(buf1, buf0, 4, tw, lane);   __syncthreads();
    fft1024_stage<INV>(buf0, buf1, 16, tw, lane);  __syncthreads();
    fft1024_stage<INV>(buf1, buf0, 64, tw, lane);  __syncthreads();
    fft1024_stage<INV>(buf0, buf1, 256, tw, lane); __syncthreads();
    return buf1;
}

// forward band kernel: grid (ceil(F/4), Q)
template <bool COMPLEX_OUT>
__global__ __launch_bounds__(256) void k_band_fwd(const float2* __restrict__ X, float* __restrict__ out,
                                                  const int4* __restrict__ bin_tab, const float* __restrict__ window,
                                                  const float2* __restrict__ tw1024, int F, int n_blocks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* tw = reinterpret_cast<float2*>(smem);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    float2* buf0 = tw + M + wave * 2 * M;
    float2* buf1 = buf0 + M;
    const long q = blockIdx.y;
    const int bin = blockIdx.x * 4 + wave;
    const bool live = bin < F;
    for (int i = tid; i < M; i += 256) tw[i] = tw1024[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) buf0[lane + 64 * i] = make_float2(0.f, 0.f);
    __syncthreads();
    if (live) {
        const int4 bt = bin_tab[bin];                 // {spec_start, pad, length, win_off}
        const float2* x = X + q * XPAD + bt.x;
        const float* w = window + bt.w;
        for (int m = lane; m < bt.z; m += 64) {
            const float2 v = x[m];
            const float g = w[m];
            buf0[bt.y + m] = make_float2(v.x * g, v.y * g);
        }
    }
    float2* res = fft1024<true>(buf0, buf1, tw, lane);
    if (!live) return;
    const float inv = 1.0f / (float)M;
    const long b = q / n_blocks, blk = q - b * n_blocks;
    const long Tt = (long)n_blocks * M;
    if (COMPLEX_OUT) {
        float2* o = reinterpret_cast<float2*>(out) + (b * F + bin) * Tt + blk * M;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float2 v = res[lane + 64 * i];
            o[lane + 64 * i] = make_float2(v.x * inv, v.y * inv);
        }
    } else {
        float* ore = out + ((b * 2 + 0) * F + bin) * Tt + blk * M;
        float* oim = out + ((b * 2 + 1) * F + bin) * Tt + blk * M;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float2 v = res[lane + 64 * i];
            ore[lane + 64 * i] = v.x * inv;
            oim[lane + 64 * i] = v.y * inv;
        }
    }
}

// inverse band kernel: FFT of each bin's 1024 frames, times the dual window, to the ragged scratch S
template <bool COMPLEX_IN>
__global__ __launch_bounds__(256) void k_band_inv(const float* __restrict__ coeffs, float2* __restrict__ S,
                                                  const int4* __restrict__ bin_tab, const float* __restrict__ dual,
                                                  const float2* __restrict__ tw1024, int F, int n_blocks, int sum_len) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* tw = reinterpret_cast<float2*>(smem);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    float2* buf0 = tw + M + wave * 2 * M;
    float2* buf1 = buf0 + M;
    const long q = blockIdx.y;
    const int bin = blockIdx.x * 4 + wave;
    const bool live = bin < F;
    for (int i = tid; i < M; i += 256) tw[i] = tw1024[i];
    const long b = q / n_blocks, blk = q - b * n_blocks;
    const long Tt = (long)n_blocks * M;
    if (live) {
        if (COMPLEX_IN) {
            const float2* c = reinterpret_cast<const float2*>(coeffs) + (b * F + bin) * Tt + blk * M;
#pragma unroll
            for (int i = 0; i < 16; ++i) buf0[lane + 64 * i] = c[lane + 64 * i];
        } else {
            const float* cre = coeffs + ((b * 2 + 0) * F + bin) * Tt + blk * M;
            const float* cim = coeffs + ((b * 2 + 1) * F + bin) * Tt + blk * M;
#pragma unroll
            for (int i = 0; i < 16; ++i) buf0[lane + 64 * i] = make_float2(cre[lane + 64 * i], cim[lane + 64 * i]);
        }
    }
    float2* res = fft1024<false>(buf0, buf1, tw, lane);
    if (!live) return;
    const int4 bt = bin_tab[bin];
    float2* s = S + q * sum_len + bt.w;
    const float* d = dual + bt.w;
    for (int m = lane; m < bt.z; m += 64) {
        const float2 v = res[bt.y + m];
        const float g = d[m];
        s[m] = make_float2(v.x * g, v.y * g);
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

__global__ __launch_bounds__(256) void k_absmax(const float* __restrict__ x, unsigned* __restrict__ out, long n) {
    float m = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));   // m >= 0: uint order == float order
}

__global__ __launch_bounds__(256) void k_scale_by_max(float* __restrict__ x, const unsigned* __restrict__ mx, long n) {
    const float m = __uint_as_float(*mx);
    if (!(m > 0.f)) return;                                            // cqtwrapper.py:209 guard
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] = x[i] / m;
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

constexpr int LDS_ROWS = (2 * ROWS * N1 + N1) * 8;    // 81,000 B
constexpr int LDS_BAND = (M + 4 * 2 * M) * 8;         // 73,728 B

}  // namespace

extern "C" int64_t tt_cqt_scratch_bytes(int n_clips, int n_bins, int sum_len) {
    (void)n_bins;
    return 256 + align256((int64_t)n_clips * NC * 8) + align256((int64_t)n_clips * XPAD * 8) +
           align256((int64_t)n_clips * sum_len * 8);
}

static int cqt_set_attrs() {
    static bool done = false;
    if (done) return 0;
    TT_HIP(hipFuncSetAttribute((const void*)k_fft675_rows, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_ROWS));
    TT_HIP(hipFuncSetAttribute((const void*)k_band_fwd<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BAND));
    TT_HIP(hipFuncSetAttribute((const void*)k_band_fwd<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BAND));
    TT_HIP(hipFuncSetAttribute((const void*)k_band_inv<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BAND));
    TT_HIP(hipFuncSetAttribute((const void*)k_band_inv<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BAND));
    done = true;
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
    hipLaunchKernelGGL(k_fft675_rows, dim3(N2 / ROWS, Q), dim3(256), LDS_ROWS, st,
                       reinterpret_cast<const float2*>(audio), s.A,
                       reinterpret_cast<const float2*>(plan->tw675), reinterpret_cast<const float2*>(plan->twNc));
    TT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_fft49_cols<0>, dim3(NTILES, Q), dim3(256), 0, st, s.A, s.X,
                       reinterpret_cast<const float2*>(plan->tw49), reinterpret_cast<const float2*>(plan->twN));
    TT_LAUNCH_CHECK();
    dim3 grid((F + 3) / 4, Q);
    if (out_complex)
        hipLaunchKernelGGL(k_band_fwd<true>, grid, dim3(256), LDS_BAND, st, s.X, out,
                           reinterpret_cast<const int4*>(plan->bin_tab), plan->window,
                           reinterpret_cast<const float2*>(plan->tw1024), F, n_blocks);
    else
        hipLaunchKernelGGL(k_band_fwd<false>, grid, dim3(256), LDS_BAND, st, s.X, out,
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
        hipLaunchKernelGGL(k_band_inv<true>, grid, dim3(256), LDS_BAND, st, coeffs, s.S,
                           reinterpret_cast<const int4*>(plan->bin_tab), plan->dual,
                           reinterpret_cast<const float2*>(plan->tw1024), F, n_blocks, plan->sum_len);
    else
        hipLaunchKernelGGL(k_band_inv<false>, grid, dim3(256), LDS_BAND, st, coeffs, s.S,
                           reinterpret_cast<const int4*>(plan->bin_tab), plan->dual,
                           reinterpret_cast<const float2*>(plan->tw1024), F, n_blocks, plan->sum_len);
    TT_LAUNCH_CHECK();
    float2* Zc = s.X;   // reuse: [Q][NC]
    hipLaunchKernelGGL(k_spec_gather, dim3((NC + 255) / 256, Q), dim3(256), 0, st, s.S, Zc, plan->gat_off,
                       plan->gat_idx, reinterpret_cast<const float2*>(plan->twN), plan->sum_len);
    TT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_fft675_rows, dim3(N2 / ROWS, Q), dim3(256), LDS_ROWS, st, Zc, s.A,
                       reinterpret_cast<const float2*>(plan->tw675), reinterpret_cast<const float2*>(plan->twNc));
    TT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_fft49_cols<1>, dim3(NTILES, Q), dim3(256), 0, st, s.A, reinterpret_cast<float2*>(audio),
                       reinterpret_cast<const float2*>(plan->tw49), reinterpret_cast<const float2*>(plan->twN));
    TT_LAUNCH_CHECK();
    if (normalize) {
        const long n = (long)Q * 2 * NC;
        TT_HIP(hipMemsetAsync(s.mx, 0, 4, st));
        hipLaunchKernelGGL(k_absmax, dim3(1024), dim3(256), 0, st, audio, s.mx, n);
        TT_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_scale_by_max, dim3(2048), dim3(256), 0, st, audio, s.mx, n);
        TT_LAUNCH_CHECK();
    }
    return 0;
}
