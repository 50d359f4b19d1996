// NSGT constant-Q transform for ANY block length (reference timbre_trap/framework/cqtwrapper.py:15-48 takes arbitrary
// secs_per_block / sample_rate; cqt_pytorch.CQT.encode / .decode as called at :67 and :207).
//
// csrc/cqt.hip is specialised for the reference configuration (N = 66150 = 2 * 675 * 49, M = 1024, the FFTs held in LDS /
// registers); this file is the general path behind the same wrapper, for every other (N, M): the constructor of the drop-in API
// must not throw for `CQT(..., secs_per_block != 3)`.  It is the SLOW path by design -- global-memory FFT passes, no fusion --
// and shares every convention of the transform with the fast one through the same host-built tables (nsgt_plan.build_plan).
//
//   length-N DFT of a block        Bluestein: X[k] = c[k] sum_n (x[n] c[n]) conj(c[k - n]),  c[n] = exp(-i pi n^2 / N), as a cyclic
//                                  convolution of length P = 2^p >= 2N - 1 (two power-of-two FFTs and a table product; any N,
//                                  prime factors included)
//   power-of-two FFTs (P and M)    Stockham autosort, radix 4 (one radix-2 pass first when log2 is odd), one launch per pass,
//                                  batched; twiddles from a host table exp(-2 pi i q / P) built in float64
//   inverse FFTs                   conj(FFT(conj z)), the conjugations folded into the neighbouring pointwise kernels
//   per-bin crop, window, layout   as in cqt.hip: bin_tab {spec_start, pad, length, win_off}, ragged window / dual tables,
//                                  CSR gather for the deterministic overlap-add; to_real / to_complex fused (planes or interleaved)
#include "common.h"

namespace {

constexpr int GT = 256;

inline int64_t align256(int64_t b) { return (b + 255) & ~(int64_t)255; }

inline int blocks_for(long n) {
    long g = (n + GT - 1) / GT;
    if (g > (1L << 20)) g = 1L << 20;
    return (int)(g < 1 ? 1 : g);
}

// ---- batched power-of-two FFT passes (src -> dst, natural order in, natural order out after all passes) ----------------------

__global__ __launch_bounds__(GT) void k_g_fft_r2(const float2* __restrict__ src, float2* __restrict__ dst, const float2* __restrict__ tw,
                                                 int logP, int s, long total) {                 // total = batch * P / 2
    const int half = 1 << (logP - 1), Ns = 1 << s;
    for (long g = (long)blockIdx.x * GT + threadIdx.x; g < total; g += (long)gridDim.x * GT) {
        const long b = g >> (logP - 1);
        const int j = (int)(g & (half - 1)), k = j & (Ns - 1);
        const float2* x = src + (b << logP);
        float2* y = dst + (b << logP);
        const float2 a = x[j], c = cmul(x[j + half], tw[(long)k << (logP - 1 - s)]);
        const int i0 = ((j - k) << 1) + k;
        y[i0] = cadd(a, c);
        y[i0 + Ns] = csub(a, c);
    }
}

__global__ __launch_bounds__(GT) void k_g_fft_r4(const float2* __restrict__ src, float2* __restrict__ dst, const float2* __restrict__ tw,
                                                 int logP, int s, long total) {                 // stages s and s + 1; total = batch * P / 4
    const int T = 1 << (logP - 2), Ns = 1 << s;
    for (long g = (long)blockIdx.x * GT + threadIdx.x; g < total; g += (long)gridDim.x * GT) {
        const long b = g >> (logP - 2);
        const int i = (int)(g & (T - 1)), k = i & (Ns - 1);
        const float2* x = src + (b << logP);
        float2* y = dst + (b << logP);
        // exp(-2 pi i k r / (4 Ns)), r = 1, 2, 3, all from the table (k r P / (4 Ns) < P / 2 for r <= 2; r = 3 may pass P / 2: -tw)
        const long q1 = (long)k << (logP - 2 - s);
        const float2 w1 = tw[q1], w2 = tw[2 * q1];
        const long q3 = 3 * q1, hp = 1L << (logP - 1);
        float2 w3 = tw[q3 >= hp ? q3 - hp : q3];
        if (q3 >= hp) w3 = make_float2(-w3.x, -w3.y);
        const float2 u0 = x[i], u1 = cmul(x[i + T], w1), u2 = cmul(x[i + 2 * T], w2), u3 = cmul(x[i + 3 * T], w3);
        const float2 v0 = cadd(u0, u2), v1 = csub(u0, u2), v2 = cadd(u1, u3), d = csub(u1, u3);
        const float2 v3 = make_float2(d.y, -d.x);                                                // -i (u1 - u3)
        const int j = ((i - k) << 2) + k;
        y[j] = cadd(v0, v2);
        y[j + Ns] = cadd(v1, v3);
        y[j + 2 * Ns] = csub(v0, v2);
        y[j + 3 * Ns] = csub(v1, v3);
    }
}

// runs the passes; on return `cur` holds the transform and `other` is free
int run_fft(float2*& cur, float2*& other, const float2* tw, int logP, long batch, hipStream_t st) {
    int s = 0;
    if (logP & 1) {
        const long total = batch << (logP - 1);
        hipLaunchKernelGGL(k_g_fft_r2, dim3(blocks_for(total)), dim3(GT), 0, st, cur, other, tw, logP, 0, total);
        TT_LAUNCH_CHECK();
        float2* t = cur; cur = other; other = t;
        s = 1;
    }
    for (; s < logP; s += 2) {
        const long total = batch << (logP - 2);
        hipLaunchKernelGGL(k_g_fft_r4, dim3(blocks_for(total)), dim3(GT), 0, st, cur, other, tw, logP, s, total);
        TT_LAUNCH_CHECK();
        float2* t = cur; cur = other; other = t;
    }
    return 0;
}

// ---- Bluestein pieces ---------------------------------------------------------------------------------------------------------

// a[q][n] = x[q][n] c[n] (n < N), 0 up to P
__global__ __launch_bounds__(GT) void k_g_chirp_real(const float* __restrict__ x, const float2* __restrict__ chirp, float2* __restrict__ a,
                                                     int N, int logP, long total) {              // total = Q * P
    const int P = 1 << logP;
    for (long g = (long)blockIdx.x * GT + threadIdx.x; g < total; g += (long)gridDim.x * GT) {
        const long q = g >> logP;
        const int n = (int)(g & (P - 1));
        float2 v = make_float2(0.f, 0.f);
        if (n < N) { const float s = x[q * N + n]; const float2 c = chirp[n]; v = make_float2(s * c.x, s * c.y); }
        a[g] = v;
    }
}

// z <- conj(z Bf)   (Bf carries the 1 / P of the inverse transform that follows)
__global__ __launch_bounds__(GT) void k_g_filter(float2* __restrict__ z, const float2* __restrict__ bf, int logP, long total) {
    const int P = 1 << logP;
    for (long g = (long)blockIdx.x * GT + threadIdx.x; g < total; g += (long)gridDim.x * GT)
        z[g] = cconj(cmul(z[g], bf[g & (P - 1)]));
}

// forward: X[q][k] = c[k] conj(W[q][k]) for k = 0 .. N / 2  (the half-spectrum the windows live on)
__global__ __launch_bounds__(GT) void k_g_spec(const float2* __restrict__ w, const float2* __restrict__ chirp, float2* __restrict__ X, int NH1,
                                               int logP, long total) {                            // total = Q * NH1
    for (long g = (long)blockIdx.x * GT + threadIdx.x; g < total; g += (long)gridDim.x * GT) {
        const long q = g / NH1;
        const int k = (int)(g - q * NH1);
        X[g] = cmul(chirp[k], cconj(w[(q << logP) + k]));
    }
}

// ---- per-bin stage --------------------------------------------------------------------------------------------------------------

// v[q][f][m] = conj(X[q][spec_start_f + (m - pad_f)] g_f[m - pad_f]) inside the window, 0 elsewhere  (conj: the M-point INVERSE FFT
// that follows runs as a forward one)
__global__ __launch_bounds__(GT) void k_g_band_gather(const float2* __restrict__ X, const int4* __restrict__ bin_tab, const float* __restrict__ window,
                                                      float2* __restrict__ v, int F, int NH1, int logM, long total) {   // total = Q * F * M
    const int M = 1 << logM;
    for (long g = (long)blockIdx.x * GT + threadIdx.x; g < total; g += (long)gridDim.x * GT) {
        const int m = (int)(g & (M - 1));
        const long qf = g >> logM;
        const long q = qf / F;
        const int f = (int)(qf - q * F);
        const int4 bt = bin_tab[f];                                   // {spec_start, pad, length, win_off}
        const int r = m - bt.y;
        float2 o = make_float2(0.f, 0.f);
        if (r >= 0 && r < bt.z) {
            const float2 s = X[q * NH1 + bt.x + r];
            const float wv = window[bt.w + r];
            o = make_float2(s.x * wv, -s.y * wv);
        }
        v[g] = o;
    }
}

// out <- conj(W) / M in the wrapper's layout: planes (B, 2, F, n_blocks M) or interleaved complex (B, 1, F, n_blocks M)
__global__ __launch_bounds__(GT) void k_g_band_out(const float2* __restrict__ w, float* __restrict__ out, int F, int n_blocks, int logM, int interleaved,
                                                   long total) {
    const int M = 1 << logM;
    const float inv = 1.f / (float)M;
    const long T = (long)n_blocks * M;
    for (long g = (long)blockIdx.x * GT + threadIdx.x; g < total; g += (long)gridDim.x * GT) {
        const int m = (int)(g & (M - 1));
        const long qf = g >> logM;
        const long q = qf / F;
        const int f = (int)(qf - q * F);
        const long b = q / n_blocks;
        const int blk = (int)(q - b * n_blocks);
        const float2 c = w[g];
        const float re = c.x * inv, im = -c.y * inv;
        const long t = (long)blk * M + m;
        if (interleaved) {
            reinterpret_cast<float2*>(out)[(b * F + f) * T + t] = make_float2(re, im);
        } else {
            out[((b * 2 + 0) * F + f) * T + t] = re;
            out[((b * 2 + 1) * F + f) * T + t] = im;
        }
    }
}

// inverse: coefficients in the wrapper's layout -> v[q][f][m]
__global__ __launch_bounds__(GT) void k_g_band_in(const float* __restrict__ coeffs, float2* __restrict__ v, int F, int n_blocks, int logM, int interleaved,
                                                  long total) {
    const int M = 1 << logM;
    const long T = (long)n_blocks * M;
    for (long g = (long)blockIdx.x * GT + threadIdx.x; g < total; g += (long)gridDim.x * GT) {
        const int m = (int)(g & (M - 1));
        const long qf = g >> logM;
        const long q = qf / F;
        const int f = (int)(qf - q * F);
        const long b = q / n_blocks;
        const int blk = (int)(q - b * n_blocks);
        const long t = (long)blk * M + m;
        v[g] = interleaved ? reinterpret_cast<const float2*>(coeffs)[(b * F + f) * T + t]
                           : make_float2(coeffs[((b * 2 + 0) * F + f) * T + t], coeffs[((b * 2 + 1) * F + f) * T + t]);
    }
}

// overlap-add in frequency through the CSR lists (fixed order: deterministic), conjugated and chirped: the input of the Bluestein
// transform that computes FFT_N(conj Xh); zero above N / 2 (the bins span the open positive half-spectrum) and up to P
__global__ __launch_bounds__(GT) void k_g_ola(const float2* __restrict__ V, const int4* __restrict__ bin_tab, const float* __restrict__ dual,
                                              const int* __restrict__ gat_off, const int* __restrict__ gat_idx, const int* __restrict__ pos_bin,
                                              const float2* __restrict__ chirp, float2* __restrict__ a, int F, int NH1, int logM, int logP, long total) {
    const int P = 1 << logP;
    for (long g = (long)blockIdx.x * GT + threadIdx.x; g < total; g += (long)gridDim.x * GT) {
        const long q = g >> logP;
        const int j = (int)(g & (P - 1));
        float2 acc = make_float2(0.f, 0.f);
        if (j < NH1) {
            for (int e = gat_off[j]; e < gat_off[j + 1]; ++e) {
                const int pos = gat_idx[e];
                const int f = pos_bin[pos];
                const int4 bt = bin_tab[f];
                const float2 s = V[((q * F + f) << logM) + bt.y + (pos - bt.w)];
                const float d = dual[pos];
                acc.x = fmaf(s.x, d, acc.x);
                acc.y = fmaf(s.y, d, acc.y);
            }
            acc = cmul(cconj(acc), chirp[j]);
        }
        a[g] = acc;
    }
}

// x[q][n] = 2 Re(c[n] conj(W[q][n])) / N, and the running abs-max of the whole batch (non-negative floats order like their bits)
__global__ __launch_bounds__(GT) void k_g_audio(const float2* __restrict__ w, const float2* __restrict__ chirp, float* __restrict__ audio, unsigned* __restrict__ peak,
                                                int N, int logP, long total) {                   // total = Q * N
    const float sc = 2.f / (float)N;
    float mx = 0.f;
    for (long g = (long)blockIdx.x * GT + threadIdx.x; g < total; g += (long)gridDim.x * GT) {
        const long q = g / N;
        const int n = (int)(g - q * N);
        const float2 c = chirp[n], z = w[(q << logP) + n];
        const float v = (c.x * z.x + c.y * z.y) * sc;                  // Re(c conj(z))
        audio[g] = v;
        mx = fmaxf(mx, fabsf(v));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0 && mx > 0.f) atomicMax(peak, __float_as_uint(mx));
}

__global__ __launch_bounds__(GT) void k_g_scale(float* __restrict__ audio, const unsigned* __restrict__ peak, long total) {
    const float p = __uint_as_float(peak[0]);
    if (!(p > 0.f)) return;                                            // cqtwrapper.py:209: only when the maximum is non-zero
    for (long g = (long)blockIdx.x * GT + threadIdx.x; g < total; g += (long)gridDim.x * GT) audio[g] = audio[g] / p;
}

struct GScratch {
    float2 *a, *b, *spec;
    unsigned* peak;
};

inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

inline bool plan_ok(const tt_cqt_gplan* p) {
    return p && p->chirp && p->bfilt && p->twP && p->twM && p->bin_tab && p->window && p->dual && p->gat_off && p->gat_idx && p->pos_bin &&
           p->n_bins > 0 && p->N >= 2 && p->M >= 2 && (p->M & (p->M - 1)) == 0 && (p->P & (p->P - 1)) == 0 && p->P >= 2 * p->N - 1 &&
           p->P <= (1 << 28);
}

inline int64_t work_elems(const tt_cqt_gplan* p, int64_t Q) {
    const int64_t band = Q * p->n_bins * p->M, blue = Q * p->P;
    return band > blue ? band : blue;
}

inline GScratch gcarve(void* scratch, const tt_cqt_gplan* p, int64_t Q) {
    char* c = reinterpret_cast<char*>(scratch);
    GScratch s;
    s.peak = reinterpret_cast<unsigned*>(c); c += 256;
    const int64_t w = align256(work_elems(p, Q) * 8);
    s.a = reinterpret_cast<float2*>(c); c += w;
    s.b = reinterpret_cast<float2*>(c); c += w;
    s.spec = reinterpret_cast<float2*>(c);
    return s;
}

}  // namespace

extern "C" int64_t tt_cqt_generic_scratch_bytes(const tt_cqt_gplan* plan, int n_clips) {
    if (!plan_ok(plan) || n_clips <= 0) return -1;
    return 256 + 2 * align256(work_elems(plan, n_clips) * 8) + align256((int64_t)n_clips * (plan->N / 2 + 1) * 8);
}

extern "C" int tt_cqt_generic_forward(const tt_cqt_gplan* plan, const float* audio, float* out, void* scratch, int B, int n_blocks,
                                      int out_complex, void* stream) {
    if (!plan_ok(plan) || !audio || !out || !scratch || B <= 0 || n_blocks <= 0) return TT_E_BADARG;
    hipStream_t st = tt_stream(stream);
    const long Q = (long)B * n_blocks;
    const int N = plan->N, NH1 = N / 2 + 1, F = plan->n_bins, logP = ilog2(plan->P), logM = ilog2(plan->M);
    GScratch s = gcarve(scratch, plan, Q);
    const float2* chirp = reinterpret_cast<const float2*>(plan->chirp);
    float2 *cur = s.a, *other = s.b;
    long total = Q << logP;
    hipLaunchKernelGGL(k_g_chirp_real, dim3(blocks_for(total)), dim3(GT), 0, st, audio, chirp, cur, N, logP, total);
    TT_LAUNCH_CHECK();
    int rc = run_fft(cur, other, reinterpret_cast<const float2*>(plan->twP), logP, Q, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_g_filter, dim3(blocks_for(total)), dim3(GT), 0, st, cur, reinterpret_cast<const float2*>(plan->bfilt), logP, total);
    TT_LAUNCH_CHECK();
    rc = run_fft(cur, other, reinterpret_cast<const float2*>(plan->twP), logP, Q, st);
    if (rc) return rc;
    total = Q * NH1;
    hipLaunchKernelGGL(k_g_spec, dim3(blocks_for(total)), dim3(GT), 0, st, cur, chirp, s.spec, NH1, logP, total);
    TT_LAUNCH_CHECK();
    cur = s.a; other = s.b;
    total = (Q * F) << logM;
    hipLaunchKernelGGL(k_g_band_gather, dim3(blocks_for(total)), dim3(GT), 0, st, s.spec, reinterpret_cast<const int4*>(plan->bin_tab), plan->window, cur, F,
                       NH1, logM, total);
    TT_LAUNCH_CHECK();
    rc = run_fft(cur, other, reinterpret_cast<const float2*>(plan->twM), logM, Q * F, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_g_band_out, dim3(blocks_for(total)), dim3(GT), 0, st, cur, out, F, n_blocks, logM, out_complex, total);
    TT_LAUNCH_CHECK();
    return 0;
}

extern "C" int tt_cqt_generic_inverse(const tt_cqt_gplan* plan, const float* coeffs, float* audio, void* scratch, int B, int n_blocks,
                                      int in_complex, int normalize, void* stream) {
    if (!plan_ok(plan) || !coeffs || !audio || !scratch || B <= 0 || n_blocks <= 0) return TT_E_BADARG;
    hipStream_t st = tt_stream(stream);
    const long Q = (long)B * n_blocks;
    const int N = plan->N, NH1 = N / 2 + 1, F = plan->n_bins, logP = ilog2(plan->P), logM = ilog2(plan->M);
    GScratch s = gcarve(scratch, plan, Q);
    const float2* chirp = reinterpret_cast<const float2*>(plan->chirp);
    float2 *cur = s.a, *other = s.b;
    long total = (Q * F) << logM;
    hipLaunchKernelGGL(k_g_band_in, dim3(blocks_for(total)), dim3(GT), 0, st, coeffs, cur, F, n_blocks, logM, in_complex, total);
    TT_LAUNCH_CHECK();
    int rc = run_fft(cur, other, reinterpret_cast<const float2*>(plan->twM), logM, Q * F, st);
    if (rc) return rc;
    total = Q << logP;
    hipLaunchKernelGGL(k_g_ola, dim3(blocks_for(total)), dim3(GT), 0, st, cur, reinterpret_cast<const int4*>(plan->bin_tab), plan->dual, plan->gat_off,
                       plan->gat_idx, plan->pos_bin, chirp, other, F, NH1, logM, logP, total);
    TT_LAUNCH_CHECK();
    { float2* t = cur; cur = other; other = t; }
    rc = run_fft(cur, other, reinterpret_cast<const float2*>(plan->twP), logP, Q, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_g_filter, dim3(blocks_for(total)), dim3(GT), 0, st, cur, reinterpret_cast<const float2*>(plan->bfilt), logP, total);
    TT_LAUNCH_CHECK();
    rc = run_fft(cur, other, reinterpret_cast<const float2*>(plan->twP), logP, Q, st);
    if (rc) return rc;
    TT_HIP(hipMemsetAsync(s.peak, 0, 4, st));
    total = Q * N;
    hipLaunchKernelGGL(k_g_audio, dim3(blocks_for(total)), dim3(GT), 0, st, cur, chirp, audio, s.peak, N, logP, total);
    TT_LAUNCH_CHECK();
    if (normalize) {
        hipLaunchKernelGGL(k_g_scale, dim3(blocks_for(total)), dim3(GT), 0, st, audio, s.peak, total);
        TT_LAUNCH_CHECK();
    }
    return 0;
}
