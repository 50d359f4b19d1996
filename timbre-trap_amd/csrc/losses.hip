// Objectives of Timbre-Trap on gfx950: reconstruction / consistency (sum of squared differences),
// TimbreTrap.to_activations, positive-class-weighted transcription loss -- forward and backward.
// Replaces reference timbre_trap/framework/objectives.py:11-104 and modules.py:271-289.
//
// All of them are pure HBM streams; reductions are two-stage in fp64 (block partials -> one
// finalising workgroup) so the loss values are deterministic.
// Also here: clip + AdamW over the flat parameter buffer, and the helpers next to the path: per-parameter
// gradient statistics, peak picking / thresholding of activations, target generation (float64, SciPy-identical).
#include "common.h"

namespace {

constexpr int MAXP = 1024;

__device__ __forceinline__ void block_partial(double v, double* partials) {
    __shared__ double red[4];
    v = wave_sum_d(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void k_finalize(const double* __restrict__ partials, int n, double scale,
                                                  float* __restrict__ out, int take_sqrt) {
    __shared__ double red[4];
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) v += partials[i];
    v = wave_sum_d(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = (red[0] + red[1] + red[2] + red[3]) * scale;
        out[0] = (float)(take_sqrt ? sqrt(s) : s);
    }
}

__global__ __launch_bounds__(256) void k_sqdiff_partial(const float* __restrict__ a, const float* __restrict__ b,
                                                        double* __restrict__ partials, long n) {
    const long n4 = n >> 2;
    const float4* a4 = reinterpret_cast<const float4*>(a);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    double acc = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 x = a4[i], y = b4[i];
        const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
        acc += (double)(d0 * d0 + d1 * d1) + (double)(d2 * d2 + d3 * d3);
    }
    if (blockIdx.x == 0) {
        const long i = (n4 << 2) + threadIdx.x;
        if (i < n) { const float d = a[i] - b[i]; acc += (double)(d * d); }
    }
    block_partial(acc, partials);
}

// Round 5: the loss and its gradient in ONE pass.  The gradient of scale * sum((a - b)^2) is 2 scale (a - b) times the incoming scalar,
// which is 1 whenever the loss goes into the total as it is (train.py:453-466): the forward pass, which reads a and b anyway, writes
// 2 scale (a - b) [and its negative / the second term's], and the backward pass is k_rescale* -- a launch whose every workgroup reads
// the incoming scalar(s) and returns when they are 1 -- instead of a second read of both operands.  Same loops and partial sums as
// k_sqdiff_partial: the loss values are bit-identical to tt_sqdiff_sum's.  TWO: both consistency terms against the same b.
template <bool TWO>
__global__ __launch_bounds__(256) void k_sqdiff_fused(const float* __restrict__ a1, const float* __restrict__ a2, const float* __restrict__ b,
                                                      double* __restrict__ partials, float s2, float* __restrict__ da1,
                                                      float* __restrict__ da2, float* __restrict__ db, long n) {
    const long n4 = n >> 2;
    const float4* p1 = reinterpret_cast<const float4*>(a1);
    const float4* p2 = reinterpret_cast<const float4*>(a2);
    const float4* pb = reinterpret_cast<const float4*>(b);
    double acc1 = 0.0, acc2 = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 y = pb[i], x = p1[i];
        const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
        acc1 += (double)(d0 * d0 + d1 * d1) + (double)(d2 * d2 + d3 * d3);
        const float4 u = float4{s2 * d0, s2 * d1, s2 * d2, s2 * d3};
        float4 v = float4{0.f, 0.f, 0.f, 0.f};
        if (TWO) {
            const float4 z = p2[i];
            const float e0 = z.x - y.x, e1 = z.y - y.y, e2 = z.z - y.z, e3 = z.w - y.w;
            acc2 += (double)(e0 * e0 + e1 * e1) + (double)(e2 * e2 + e3 * e3);
            v = float4{s2 * e0, s2 * e1, s2 * e2, s2 * e3};
            if (da2) reinterpret_cast<float4*>(da2)[i] = v;
        }
        if (da1) reinterpret_cast<float4*>(da1)[i] = u;
        if (db) reinterpret_cast<float4*>(db)[i] = float4{-u.x - v.x, -u.y - v.y, -u.z - v.z, -u.w - v.w};
    }
    if (blockIdx.x == 0) {
        const long i = (n4 << 2) + threadIdx.x;
        if (i < n) {
            const float d = a1[i] - b[i];
            acc1 += (double)(d * d);
            float v = 0.f;
            if (TWO) { const float e = a2[i] - b[i]; acc2 += (double)(e * e); v = s2 * e; if (da2) da2[i] = v; }
            if (da1) da1[i] = s2 * d;
            if (db) db[i] = -(s2 * d) - v;
        }
    }
    block_partial(acc1, partials);
    if (TWO) { __syncthreads(); block_partial(acc2, partials + MAXP); }
}

// t *= g (and tneg, the stored negative); nothing when g == 1 (every workgroup reads the scalar and returns)
__global__ __launch_bounds__(256) void k_rescale1(float* __restrict__ t, float* __restrict__ tneg, const float* __restrict__ g, long n) {
    const float gv = g[0];
    if (gv == 1.f) return;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        if (t) t[i] *= gv;
        if (tneg) tneg[i] *= gv;
    }
}
// da1 *= g1, da2 *= g2, db = -(g1 u + g2 v) from the stored unit-scale gradients u, v (db alone cannot be rescaled: it is their sum); a
// missing incoming gradient counts as 0.  Needs da1 and da2 (the caller keeps them even when only db is wanted).
__global__ __launch_bounds__(256) void k_rescale2(float* __restrict__ da1, float* __restrict__ da2, float* __restrict__ db,
                                                  const float* __restrict__ g1, const float* __restrict__ g2, long n) {
    const float s1 = g1 ? g1[0] : 0.f, s2 = g2 ? g2[0] : 0.f;
    if (s1 == 1.f && s2 == 1.f) return;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float u = s1 * da1[i], v = s2 * da2[i];
        da1[i] = u; da2[i] = v;
        if (db) db[i] = -u - v;
    }
}

__global__ __launch_bounds__(256) void k_sqdiff_bwd(const float* __restrict__ a, const float* __restrict__ b,
                                                    const float* __restrict__ gscale, float scale,
                                                    float* __restrict__ da, float* __restrict__ db, long n) {
    const float s = 2.f * scale * gscale[0];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float g = s * (a[i] - b[i]);
        if (da) da[i] = g;
        if (db) db[i] = -g;
    }
}

__global__ __launch_bounds__(256) void k_act_fwd(const float* __restrict__ c, float* __restrict__ act, int B, long FT) {
    const long total = (long)B * FT;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long b = i / FT, r = i - b * FT;
        const float re = c[(b * 2) * FT + r], im = c[(b * 2 + 1) * FT + r];
        act[i] = tanhf(sqrtf(re * re + im * im));
    }
}

__global__ __launch_bounds__(256) void k_act_bwd(const float* __restrict__ c, const float* __restrict__ act,
                                                 const float* __restrict__ dact, float* __restrict__ dc, int B, long FT) {
    const long total = (long)B * FT;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long b = i / FT, r = i - b * FT;
        const float re = c[(b * 2) * FT + r], im = c[(b * 2 + 1) * FT + r];
        const float mag = sqrtf(re * re + im * im);
        const float a = act[i];
        // d tanh = 1 - a^2 ; d|c|/d re = re / |c| (0 at the origin, torch.norm's subgradient)
        const float k = (mag > 0.f) ? dact[i] * (1.f - a * a) / mag : 0.f;
        dc[(b * 2) * FT + r] = k * re;
        dc[(b * 2 + 1) * FT + r] = k * im;
    }
}

// A workgroup = 64 consecutive frames x 4 quarters of the F bins (wave = quarter, so every load is 256 contiguous bytes); the
// quarters meet in LDS in a fixed order.  (One thread per frame looping all 540 bins twice left the launch at 65,536 threads of
// dependent loads: 0.43 ms for 0.42 GB; round 3.)
// dest != NULL (round 5, like k_sqdiff_fused): the gradient w.r.t. the estimate for an incoming scalar of 1, s1 w (est - tgt) with
// s1 = 2 / (B T), written in the same pass (k_trn_bwd's expression); backward is then k_rescale1
__global__ __launch_bounds__(256) void k_trn_fwd(const float* __restrict__ est, const float* __restrict__ tgt,
                                                 float* __restrict__ frame_scale, double* __restrict__ partials,
                                                 int B, int F, int T, int weighted, float* __restrict__ dest = nullptr, float s1 = 0.f) {
    __shared__ float sp[4][64], sn[4][64], ss[4][64];
    const long total = (long)B * T;
    const int tl = threadIdx.x & 63, fq = threadIdx.x >> 6;
    const int f0 = (int)((long)F * fq / 4), f1 = (int)((long)F * (fq + 1) / 4);
    const long nblk = (total + 63) / 64;
    double acc = 0.0;
    for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const long i = blk * 64 + tl;
        const bool valid = i < total;
        const long ic = valid ? i : total - 1;
        const long b = ic / T, t = ic - b * T;
        const float* e = est + b * F * (long)T + t;
        const float* g = tgt + b * F * (long)T + t;
        float scale = 1.f;
        if (weighted) {
            float pos = 0.f, neg = 0.f;
            for (int f = f0; f < f1; ++f) { const float v = g[(long)f * T]; pos += v; neg += 1.f - v; }
            sp[fq][tl] = pos; sn[fq][tl] = neg;
            __syncthreads();
            pos = (sp[0][tl] + sp[1][tl]) + (sp[2][tl] + sp[3][tl]);
            neg = (sn[0][tl] + sn[1][tl]) + (sn[2][tl] + sn[3][tl]);
            scale = neg / (pos + 1.1920928955078125e-07f);     // torch.finfo(float32).eps
            if (fq == 0 && valid) frame_scale[i] = scale;
        }
        float s = 0.f;
        for (int f = f0; f < f1; ++f) {
            const float tv = g[(long)f * T];
            const float d = e[(long)f * T] - tv;
            float w = 1.f;
            if (weighted && tv == 1.f && scale != 0.f) w = scale;
            s = fmaf(d * d, w, s);
            if (dest && valid) dest[b * F * (long)T + t + (long)f * T] = s1 * w * d;
        }
        ss[fq][tl] = s;
        __syncthreads();
        if (fq == 0 && valid) acc += (double)((ss[0][tl] + ss[1][tl]) + (ss[2][tl] + ss[3][tl]));
        __syncthreads();                                         // the three arrays are rewritten by the next block of frames
    }
    block_partial(acc, partials);
}

__global__ __launch_bounds__(256) void k_trn_bwd(const float* __restrict__ est, const float* __restrict__ tgt,
                                                 const float* __restrict__ frame_scale, const float* __restrict__ gscale,
                                                 float* __restrict__ dest, int B, int F, int T, int weighted) {
    const long total = (long)B * F * T;
    const float s = 2.f * gscale[0] / (float)((long)B * T);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long b = i / ((long)F * T), t = i % T;
        const float tv = tgt[i];
        float w = 1.f;
        if (weighted && tv == 1.f) { const float sc = frame_scale[b * T + t]; if (sc != 0.f) w = sc; }
        dest[i] = s * w * (est[i] - tv);
    }
}

inline int nblocks(long n, int per_thread) {
    long g = (n + 256L * per_thread - 1) / (256L * per_thread);
    if (g > MAXP) g = MAXP;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" int tt_sqdiff_sum(const float* a, const float* b, float* loss, double* partials, int64_t n,
                             float scale, void* stream) {
    if (!a || !b || !loss || !partials || n <= 0) return TT_E_BADARG;
    const int g = nblocks(n, 16);
    hipLaunchKernelGGL(k_sqdiff_partial, dim3(g), dim3(256), 0, tt_stream(stream), a, b, partials, (long)n);
    TT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, tt_stream(stream), partials, g, (double)scale, loss, 0);
    TT_LAUNCH_CHECK();
    return 0;
}

extern "C" int tt_sqdiff_sum_grad(const float* a, const float* b, float* loss, double* partials, int64_t n, float scale, float* da,
                                  float* db, void* stream) {
    if (!a || !b || !loss || !partials || n <= 0) return TT_E_BADARG;
    if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)da | (uintptr_t)db) & 15) return TT_E_BADARG;
    const int g = nblocks(n, 16);
    hipLaunchKernelGGL(k_sqdiff_fused<false>, dim3(g), dim3(256), 0, tt_stream(stream), a, (const float*)nullptr, b, partials, 2.f * scale, da,
                       (float*)nullptr, db, (long)n);
    TT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, tt_stream(stream), partials, g, (double)scale, loss, 0);
    TT_LAUNCH_CHECK();
    return 0;
}

extern "C" int tt_sqdiff2_sum_grad(const float* a1, const float* a2, const float* b, float* l1, float* l2, double* partials, int64_t n,
                                   float scale, float* da1, float* da2, float* db, void* stream) {
    if (!a1 || !a2 || !b || !l1 || !l2 || !partials || n <= 0) return TT_E_BADARG;
    if (((uintptr_t)a1 | (uintptr_t)a2 | (uintptr_t)b | (uintptr_t)da1 | (uintptr_t)da2 | (uintptr_t)db) & 15) return TT_E_BADARG;
    const int g = nblocks(n, 16);
    hipLaunchKernelGGL(k_sqdiff_fused<true>, dim3(g), dim3(256), 0, tt_stream(stream), a1, a2, b, partials, 2.f * scale, da1, da2, db, (long)n);
    TT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, tt_stream(stream), partials, g, (double)scale, l1, 0);
    TT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, tt_stream(stream), partials + MAXP, g, (double)scale, l2, 0);
    TT_LAUNCH_CHECK();
    return 0;
}

extern "C" int tt_sqdiff_rescale(float* da, float* db, const float* g, int64_t n, void* stream) {
    if ((!da && !db) || !g || n <= 0) return TT_E_BADARG;
    hipLaunchKernelGGL(k_rescale1, dim3(nblocks(n, 4) * 4), dim3(256), 0, tt_stream(stream), da, db, g, (long)n);
    TT_LAUNCH_CHECK();
    return 0;
}

extern "C" int tt_sqdiff2_rescale(float* da1, float* da2, float* db, const float* g1, const float* g2, int64_t n, void* stream) {
    if (!da1 || !da2 || n <= 0) return TT_E_BADARG;
    hipLaunchKernelGGL(k_rescale2, dim3(nblocks(n, 4) * 4), dim3(256), 0, tt_stream(stream), da1, da2, db, g1, g2, (long)n);
    TT_LAUNCH_CHECK();
    return 0;
}

// Two squared-error terms against the SAME second operand (the consistency losses, reference objectives.py:77-104: both reconstruct
// the non-detached transcription coefficients): da1 = 2 s g1 (a1 - b), da2 = 2 s g2 (a2 - b), db = -(da1 + da2) in one pass -- the two
// tt_sqdiff_bwd calls wrote db twice and left the sum to a third elementwise kernel (11 tensor passes instead of 6).
__global__ __launch_bounds__(256) void k_sqdiff2_bwd(const float4* __restrict__ a1, const float4* __restrict__ a2, const float4* __restrict__ b,
                                                     const float* __restrict__ g1, const float* __restrict__ g2, float scale,
                                                     float4* __restrict__ da1, float4* __restrict__ da2, float4* __restrict__ db, long n4,
                                                     long n) {
    const float s1 = g1 ? 2.f * scale * g1[0] : 0.f, s2 = g2 ? 2.f * scale * g2[0] : 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 x1 = a1[i], x2 = a2[i], y = b[i];
        const float4 u = float4{s1 * (x1.x - y.x), s1 * (x1.y - y.y), s1 * (x1.z - y.z), s1 * (x1.w - y.w)};
        const float4 v = float4{s2 * (x2.x - y.x), s2 * (x2.y - y.y), s2 * (x2.z - y.z), s2 * (x2.w - y.w)};
        if (da1) da1[i] = u;
        if (da2) da2[i] = v;
        if (db) db[i] = float4{-u.x - v.x, -u.y - v.y, -u.z - v.z, -u.w - v.w};
    }
    if (blockIdx.x == 0) {                                       // tail of a length that is not a multiple of four
        const long i = (n4 << 2) + threadIdx.x;
        if (i < n) {
            const float* p1 = reinterpret_cast<const float*>(a1); const float* p2 = reinterpret_cast<const float*>(a2);
            const float* pb = reinterpret_cast<const float*>(b);
            const float u = s1 * (p1[i] - pb[i]), v = s2 * (p2[i] - pb[i]);
            if (da1) reinterpret_cast<float*>(da1)[i] = u;
            if (da2) reinterpret_cast<float*>(da2)[i] = v;
            if (db) reinterpret_cast<float*>(db)[i] = -u - v;
        }
    }
}

extern "C" int tt_sqdiff2_bwd(const float* a1, const float* a2, const float* b, const float* g1, const float* g2, float scale, float* da1,
                              float* da2, float* db, int64_t n, void* stream) {
    if (!a1 || !a2 || !b || n <= 0) return TT_E_BADARG;
    const uintptr_t al = (uintptr_t)a1 | (uintptr_t)a2 | (uintptr_t)b | (uintptr_t)da1 | (uintptr_t)da2 | (uintptr_t)db;
    if (al & 15) return TT_E_BADARG;                             // 16-byte accesses (torch allocations are 256-byte aligned)
    hipLaunchKernelGGL(k_sqdiff2_bwd, dim3(nblocks(n >> 2, 4) * 4), dim3(256), 0, tt_stream(stream), (const float4*)a1, (const float4*)a2,
                       (const float4*)b, g1, g2, scale, (float4*)da1, (float4*)da2, (float4*)db, (long)(n >> 2), (long)n);
    TT_LAUNCH_CHECK();
    return 0;
}

extern "C" int tt_sqdiff_bwd(const float* a, const float* b, const float* gscale, float scale, float* da,
                             float* db, int64_t n, void* stream) {
    if (!a || !b || !gscale || n <= 0 || (!da && !db)) return TT_E_BADARG;
    hipLaunchKernelGGL(k_sqdiff_bwd, dim3(nblocks(n, 4) * 4), dim3(256), 0, tt_stream(stream), a, b, gscale, scale, da, db, (long)n);
    TT_LAUNCH_CHECK();
    return 0;
}

extern "C" int tt_activations_fwd(const float* coeffs, float* act, int B, int F, int T, void* stream) {
    if (!coeffs || !act || B <= 0 || F <= 0 || T <= 0) return TT_E_BADARG;
    const long FT = (long)F * T;
    hipLaunchKernelGGL(k_act_fwd, dim3(nblocks(B * FT, 4) * 4), dim3(256), 0, tt_stream(stream), coeffs, act, B, FT);
    TT_LAUNCH_CHECK();
    return 0;
}

extern "C" int tt_activations_bwd(const float* coeffs, const float* act, const float* dact, float* dcoeffs,
                                  int B, int F, int T, void* stream) {
    if (!coeffs || !act || !dact || !dcoeffs || B <= 0 || F <= 0 || T <= 0) return TT_E_BADARG;
    const long FT = (long)F * T;
    hipLaunchKernelGGL(k_act_bwd, dim3(nblocks(B * FT, 4) * 4), dim3(256), 0, tt_stream(stream), coeffs, act, dact, dcoeffs, B, FT);
    TT_LAUNCH_CHECK();
    return 0;
}

extern "C" int tt_transcription_loss_fwd(const float* est, const float* tgt, float* loss, float* frame_scale,
                                         double* partials, int B, int F, int T, int weighted, void* stream) {
    if (!est || !tgt || !loss || !partials || B <= 0 || F <= 0 || T <= 0) return TT_E_BADARG;
    if (weighted && !frame_scale) return TT_E_BADARG;
    const int g = nblocks((long)B * T, 1) * 4 > MAXP ? MAXP : nblocks((long)B * T, 1) * 4;     // 64 frames per workgroup
    hipLaunchKernelGGL(k_trn_fwd, dim3(g), dim3(256), 0, tt_stream(stream), est, tgt, frame_scale, partials, B, F, T, weighted);
    TT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, tt_stream(stream), partials, g, 1.0 / ((double)B * T), loss, 0);
    TT_LAUNCH_CHECK();
    return 0;
}

extern "C" int tt_transcription_loss_fwd_grad(const float* est, const float* tgt, float* loss, float* frame_scale, double* partials,
                                              float* dest, int B, int F, int T, int weighted, void* stream) {
    if (!est || !tgt || !loss || !partials || !dest || B <= 0 || F <= 0 || T <= 0) return TT_E_BADARG;
    if (weighted && !frame_scale) return TT_E_BADARG;
    const int g = nblocks((long)B * T, 1) * 4 > MAXP ? MAXP : nblocks((long)B * T, 1) * 4;
    hipLaunchKernelGGL(k_trn_fwd, dim3(g), dim3(256), 0, tt_stream(stream), est, tgt, frame_scale, partials, B, F, T, weighted, dest,
                       2.f / (float)((long)B * T));
    TT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, tt_stream(stream), partials, g, 1.0 / ((double)B * T), loss, 0);
    TT_LAUNCH_CHECK();
    return 0;
}

extern "C" int tt_transcription_loss_bwd(const float* est, const float* tgt, const float* frame_scale,
                                         const float* gscale, float* dest, int B, int F, int T, int weighted,
                                         void* stream) {
    if (!est || !tgt || !gscale || !dest || B <= 0 || F <= 0 || T <= 0) return TT_E_BADARG;
    if (weighted && !frame_scale) return TT_E_BADARG;
    hipLaunchKernelGGL(k_trn_bwd, dim3(nblocks((long)B * F * T, 4) * 4), dim3(256), 0, tt_stream(stream), est, tgt, frame_scale,
                       gscale, dest, B, F, T, weighted);
    TT_LAUNCH_CHECK();
    return 0;
}

// ---- optimiser (experiments/train.py:334, :493-496) ------------------------------------------
namespace {

__global__ __launch_bounds__(256) void k_sumsq_partial(const float* __restrict__ x, double* __restrict__ partials, long n) {
    double acc = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = x[i];
        acc += (double)v * (double)v;
    }
    block_partial(acc, partials);
}

__global__ __launch_bounds__(256) void k_adamw(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                               float* __restrict__ v, const float* __restrict__ norm, long n, float lr,
                                               float beta1, float beta2, float eps, float wd, int step, int* __restrict__ skipped,
                                               float max_norm, int write_clipped) {
    float clip = 1.f;
    if (norm) {
        const float nv = norm[0];
        if (skipped && !(fabsf(nv) <= 3.4e38f)) {                // inf / NaN gradient norm (an overflowed fp16 backward): skip the update,
            if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(skipped, 1);   // as torch.amp.GradScaler would; nobody READS the count on this path
            return;
        }
        if (max_norm > 0.f) {
            clip = max_norm / (nv + 1e-6f);
            clip = clip > 1.f ? 1.f : clip;
        }
    }
    // bias corrections at the number of APPLIED updates: the host's call count minus the skipped ones
    // (in double: 1 - beta2^t cancels to ~1e-3 at t = 1, where a one-ulp error of a float pow would be 6e-5 of the result)
    const double tstep = (double)(step - (skipped ? skipped[0] : 0));
    const float bc1 = (float)(1.0 - pow((double)beta1, tstep)), bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, tstep));
    const float step_size = lr / bc1;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gi = g[i] * clip;
        if (write_clipped) g[i] = gi;
        float pi = p[i] * (1.f - lr * wd);
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - step_size * (mi / denom);
    }
}

}  // namespace

extern "C" int tt_l2norm(const float* x, float* norm_out, double* partials, int64_t n, void* stream) {
    if (!x || !norm_out || !partials || n <= 0) return TT_E_BADARG;
    const int g = nblocks(n, 8);
    hipLaunchKernelGGL(k_sumsq_partial, dim3(g), dim3(256), 0, tt_stream(stream), x, partials, (long)n);
    TT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, tt_stream(stream), partials, g, 1.0, norm_out, 1);
    TT_LAUNCH_CHECK();
    return 0;
}

extern "C" int tt_adamw_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, const float* norm,
                             int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                             int step, float max_norm, int write_clipped, int32_t* skipped, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || step <= 0) return TT_E_BADARG;
    hipLaunchKernelGGL(k_adamw, dim3(nblocks(n, 4) * 4 > 2048 ? 2048 : nblocks(n, 4) * 4), dim3(256), 0, tt_stream(stream),
                       param, grad, exp_avg, exp_avg_sq, norm, (long)n, lr, beta1, beta2, eps, weight_decay, step, skipped,
                       max_norm, write_clipped);
    TT_LAUNCH_CHECK();
    return 0;
}

// ---- host-side helpers next to the path (f1 gradient statistics, f3 peak picking) ---------------------
namespace {

// one workgroup per segment [off[2s], off[2s+1]) of a flat buffer: out[2s] = L2 norm, out[2s+1] = max |x|
__global__ __launch_bounds__(256) void k_segment_stats(const float* __restrict__ x, const long* __restrict__ off,
                                                       float* __restrict__ out) {
    __shared__ double rs[4];
    __shared__ float rm[4];
    const int s = blockIdx.x;
    const long a = off[2 * s], b = off[2 * s + 1];
    double acc = 0.0;
    float mx = 0.f;
    for (long i = a + threadIdx.x; i < b; i += 256) {
        const float v = x[i];
        acc += (double)v * (double)v;
        mx = fmaxf(mx, fabsf(v));
    }
    acc = wave_sum_d(acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = acc; rm[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[2 * s] = (float)sqrt(rs[0] + rs[1] + rs[2] + rs[3]);
        out[2 * s + 1] = fmaxf(fmaxf(rm[0], rm[1]), fmaxf(rm[2], rm[3]));
    }
}

// x viewed as (n_outer, F, T): keep strict local maxima along F (zero rows assumed beyond both ends);
// mode 0: out = peak ? x : 0 ; mode 1: out = (x >= thr) ; mode 2: out = (peak && x >= thr)
__global__ __launch_bounds__(256) void k_peak_pick(const float* __restrict__ x, float* __restrict__ out, long n_outer, int F,
                                                   int T, double thr, int mode, int f_valid) {
    const long total = n_outer * F * (long)T;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int t = (int)(i % T);
        const int f = (int)((i / T) % F);
        (void)t;
        // rows f >= f_valid (bins above the evaluation range, reference experiments/evaluate.py:46,107-112) read as zero
        const float v = f < f_valid ? x[i] : 0.f;
        bool peak = true;
        if (mode != 1) {
            const float up = f > 0 ? x[i - T] : 0.f, dn = (f < F - 1 && f + 1 < f_valid) ? x[i + T] : 0.f;
            peak = v > up && v > dn;
        }
        float r;
        if (mode == 0) r = peak ? v : 0.f;
        else if (mode == 1) r = (double)v >= thr ? 1.f : 0.f;          // compared in double like the reference's ndarray >= t
        else r = (peak && (double)v >= thr) ? 1.f : 0.f;
        out[i] = r;
    }
}

}  // namespace

// ---- target generation (PitchDataset.multi_pitch_to_activations, reference PitchDataset.py:233-307) ----
// float64 like the reference; the blur follows SciPy's symmetric correlate1d order (centre, then pairs from the outside
// in) with separate multiplies and adds, so the result is bit-identical to scipy.ndimage.gaussian_filter1d.
__global__ __launch_bounds__(256) void k_tgt_scatter(const int* __restrict__ bins, const int* __restrict__ frames, int n, int T,
                                                     double* __restrict__ a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[(long)bins[i] * T + frames[i]] = 1.0;
}
__global__ __launch_bounds__(256) void k_tgt_blur(const double* __restrict__ a, const double* __restrict__ w, int r, int F, int T,
                                                  double* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)F * T) return;
    const int f = (int)(i / T), t = (int)(i - (long)f * T);
    double acc = __dmul_rn(a[i], w[r]);
    for (int j = -r; j < 0; ++j) {
        const int fl = f + j, fh = f - j;
        const double lo = fl >= 0 ? a[(long)fl * T + t] : 0.0, hi = fh < F ? a[(long)fh * T + t] : 0.0;
        acc = __dadd_rn(acc, __dmul_rn(__dadd_rn(lo, hi), w[r + j]));
    }
    out[i] = acc;
}
// smallest blurred value over the annotated (bin, frame) positions: one workgroup
__global__ __launch_bounds__(256) void k_tgt_seed_min(const int* __restrict__ bins, const int* __restrict__ frames, int n, int T,
                                                      const double* __restrict__ b, double* __restrict__ minv) {
    __shared__ double red[256];
    double m = 1.0e300;
    for (int i = threadIdx.x; i < n; i += 256) m = fmin(m, b[(long)bins[i] * T + frames[i]]);
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmin(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) *minv = red[0];
}
__global__ __launch_bounds__(256) void k_tgt_normalise(double* __restrict__ b, long n, const double* __restrict__ minv) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double v = b[i] / *minv;
    b[i] = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
}

extern "C" int tt_target_activations(const int* bins, const int* frames, int n, const double* weights, int radius, int F, int T,
                                     double* work, double* out, void* stream) {
    if (!out || F <= 0 || T <= 0 || n < 0 || radius < 0 || (n > 0 && (!bins || !frames)) || (radius > 0 && (!weights || !work)))
        return TT_E_BADARG;
    hipStream_t st = tt_stream(stream);
    const long total = (long)F * T;
    double* seeds = radius > 0 ? work : out;
    TT_HIP(hipMemsetAsync(seeds, 0, total * sizeof(double), st));
    if (n == 0) {
        if (radius > 0) TT_HIP(hipMemsetAsync(out, 0, total * sizeof(double), st));
        return 0;
    }
    hipLaunchKernelGGL(k_tgt_scatter, dim3((n + 255) / 256), dim3(256), 0, st, bins, frames, n, T, seeds);
    TT_LAUNCH_CHECK();
    if (radius == 0) return 0;
    hipLaunchKernelGGL(k_tgt_blur, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const double*)work, weights, radius, F, T, out);
    TT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_tgt_seed_min, dim3(1), dim3(256), 0, st, bins, frames, n, T, (const double*)out, work);   // work[0] is free now
    TT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_tgt_normalise, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, out, total, (const double*)work);
    TT_LAUNCH_CHECK();
    return 0;
}

extern "C" int tt_segment_stats(const float* x, const int64_t* offsets, int n_segments, float* out, void* stream) {
    if (!x || !offsets || !out || n_segments <= 0) return TT_E_BADARG;
    hipLaunchKernelGGL(k_segment_stats, dim3(n_segments), dim3(256), 0, tt_stream(stream), x, reinterpret_cast<const long*>(offsets), out);
    TT_LAUNCH_CHECK();
    return 0;
}

extern "C" int tt_peak_pick(const float* x, float* out, int64_t n_outer, int F, int T, double threshold, int mode, int f_valid,
                            void* stream) {
    if (!x || !out || n_outer <= 0 || F <= 0 || T <= 0 || mode < 0 || mode > 2) return TT_E_BADARG;
    hipLaunchKernelGGL(k_peak_pick, dim3(nblocks(n_outer * F * (long)T, 4) * 4), dim3(256), 0, tt_stream(stream), x, out, (long)n_outer,
                       F, T, threshold, mode, f_valid > 0 && f_valid < F ? f_valid : F);
    TT_LAUNCH_CHECK();
    return 0;
}

thread_local float g_tt_loss_scale = 1.0f;
extern "C" float tt_set_loss_scale(float scale) {
    const float prev = g_tt_loss_scale;
    if (scale > 0.f && scale <= 16777216.f) g_tt_loss_scale = scale;
    return prev;
}

int g_tt_cu_limit = 256;
extern "C" int tt_set_cu_limit(int cus) {
    const int prev = g_tt_cu_limit;
    if (cus > 0) g_tt_cu_limit = cus > 256 ? 256 : cus;
    return prev;
}

extern "C" int tt_version(void) { return 6; }   // 6: tt_skip_join16_{fwd,bwd} (the skip joins of the 16-bit path in one pass each way)   // 5: gate links (tt_wide_level_bwd_gated, tt_*_bwd_pregated, tt_latent16_*_{gated,pregated}, tt_gate16)   // 4: any-block-length CQT, tt_set_loss_scale, tt_adamw_step(skipped)   // 3: bf16 channels-last entry points (tt_wide_*, tt_sconv16_*, tt_tconv16_*, tt_latent16_*, tt_conv{in,out}16_*)
extern "C" const char* tt_arch(void) { return "gfx950"; }
extern "C" const char* tt_error_string(int code) {
    if (code == 0) return "ok";
    if (code == TT_E_BADARG) return "ttrap: bad argument";
    if (code == TT_E_UNSUPPORTED) return "ttrap: unsupported shape or option";
    return hipGetErrorString((hipError_t)code);
}
