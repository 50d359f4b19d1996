// Batched fp32 GEMM for the (31,1) latent layers: Encoder.convlat (reference modules.py:446) and
// Decoder.convin (modules.py:534), forward, data- and weight-gradient.  With H collapsed these are
// plain matrix products over (C*31): K = 1984 for the encoder head, M = 1984 for the decoder head.
//
// LDS-tiled 64x64x16, 256 threads, 4x4 outputs per thread, generic operand strides.
#include <cstdlib>
#include "common.h"

namespace {

struct GemmP {
    const float* A; const float* B; float* C; const float* bias;
    int M, N, K;
    long am, ak, bk, bn, ldc;       // element strides: a(m,k) = A[m*am + k*ak], b(k,n) = B[k*bk + n*bn]
    long sa, sb, sc;
    int reduce_batch;
    int sc_batches;                  // number of batches (matrix-core kernel)
    float alpha, beta;
    int bias_mode, bias_div, act;
};

constexpr int BM = 64, BN = 64, BK = 16;

__global__ __launch_bounds__(256) void k_gemm(GemmP p) {
    __shared__ float As[BK][BM + 4];
    __shared__ float Bs[BK][BN + 4];
    const int tid = threadIdx.x;
    const int bz = blockIdx.z;
    const float* A = p.A + bz * p.sa;
    const float* B = p.B + bz * p.sb;
    float* C = p.C + (p.reduce_batch ? 0 : bz * p.sc);
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int tx = tid & 15, ty = tid >> 4;      // thread tile: rows ty*4.., cols tx*4..
    float acc[4][4] = {};

    for (int k0 = 0; k0 < p.K; k0 += BK) {
        // A tile: BM x BK, 1024 elements, 4 per thread; iterate the unit-stride dimension fastest
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            int m, k;
            if (p.ak == 1) { k = e & (BK - 1); m = e >> 4; } else { m = e & (BM - 1); k = e >> 6; }
            const int gm = m0 + m, gk = k0 + k;
            As[k][m] = (gm < p.M && gk < p.K) ? A[gm * p.am + gk * p.ak] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            int n, k;
            if (p.bn == 1) { n = e & (BN - 1); k = e >> 6; } else { k = e & (BK - 1); n = e >> 4; }
            const int gn = n0 + n, gk = k0 + k;
            Bs[k][n] = (gn < p.N && gk < p.K) ? B[gk * p.bk + gn * p.bn] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < BK; ++k) {
            const float4 a = *reinterpret_cast<const float4*>(&As[k][ty * 4]);
            const float4 b = *reinterpret_cast<const float4*>(&Bs[k][tx * 4]);
            const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gm = m0 + ty * 4 + i;
        if (gm >= p.M) continue;
        float bv = 0.f;
        if (p.bias_mode == 1) bv = p.bias[gm];
        else if (p.bias_mode == 2) bv = p.bias[gm / p.bias_div];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gn = n0 + tx * 4 + j;
            if (gn >= p.N) continue;
            float v = p.alpha * acc[i][j];
            float* c = C + gm * p.ldc + gn;
            if (p.reduce_batch) {
                atomicAdd(c, v);
            } else {
                v += bv;
                if (p.beta != 0.f) v += p.beta * (*c);
                if (p.act == TT_ACT_ELU) v = elu1(v);
                *c = v;
            }
        }
    }
}

// ---- fp32 matrix-core version ---------------------------------------------------------------------------------------
// v_mfma_f32_16x16x4_f32 (exact fp32).  Workgroup = 4 waves = BM x 128 outputs, wave w owns columns 32 w .. 32 w + 31
// (BM / 16 x 2 accumulator tiles; 16-row tiles entirely past M are skipped).  BK = 16 per stage: the next stage's operands are loaded into registers while the
// current one is multiplied (one barrier pair per stage), generic element strides as above.  Pitches are 17 mod 32.  reduce_batch: a workgroup sums `bper`
// consecutive batches in registers before its atomics.
// (Measured and dropped in round 2: 32 k per stage with two LDS buffers and one barrier per stage -- 59 KB of LDS, two workgroups
// per CU instead of five -- took the latent-head family from 11.2 to 13.8 ms per train step.  Like the weight-gradient kernel,
// this one is carried by many independent workgroups per CU, not by a deeper pipeline inside one.  The same 32-k stage with a SINGLE buffer
// (29.7 KB, occupancy kept) also lost: encoder head forward 0.464 vs 0.386 ms.)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int BM_>
__global__ __launch_bounds__(256) void k_gemm_mfma(GemmP p, int bper) {
    constexpr int BN_ = 128, BK_ = 16, MT = BM_ / 16;
    constexpr int AP = BM_ + 20, BP = BN_ + 20;      // 20 mod 32 and a multiple of 4: fragment reads and k-fastest staging writes stay spread
                                                     // over the banks, and rows start 16-byte aligned for the float4 staging of unit-stride operands
    constexpr int NA = BM_ * BK_ / 256, NB = BN_ * BK_ / 256;
    __shared__ __attribute__((aligned(16))) float As[BK_ * AP];
    __shared__ __attribute__((aligned(16))) float Bs[BK_ * BP];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int m0 = blockIdx.y * BM_, n0 = blockIdx.x * BN_;
    const int b_lo = blockIdx.z * bper, b_hi = (b_lo + bper < (int)p.sc_batches) ? b_lo + bper : (int)p.sc_batches;
    f32x4 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { acc[mt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[mt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int mt_used = (p.M - m0 + 15) / 16;            // row tiles of this workgroup that hold valid rows (uniform)
    const int nk = (p.K + BK_ - 1) / BK_;
    const int nstage = nk * (b_hi - b_lo);
    // operands whose m / n index is the unit-stride one are staged 16 bytes per lane
    const bool avec = p.am == 1 && (p.ak & 3) == 0 && (p.sa & 3) == 0 && (reinterpret_cast<uintptr_t>(p.A) & 15) == 0 && m0 + BM_ <= p.M;
    const bool bvec = p.bn == 1 && (p.bk & 3) == 0 && (p.sb & 3) == 0 && (reinterpret_cast<uintptr_t>(p.B) & 15) == 0 && n0 + BN_ <= p.N;
    // k-contiguous operands (weight-gradient form): 16 bytes along k per lane, transposed on the LDS write
    const bool akvec = p.ak == 1 && (p.am & 3) == 0 && (p.sa & 3) == 0 && (reinterpret_cast<uintptr_t>(p.A) & 15) == 0 && m0 + BM_ <= p.M && (p.K & 15) == 0;
    const bool bkvec = p.bk == 1 && (p.bn & 3) == 0 && (p.sb & 3) == 0 && (reinterpret_cast<uintptr_t>(p.B) & 15) == 0 && n0 + BN_ <= p.N && (p.K & 15) == 0;
    float ra[NA], rb[NB];
    auto load = [&](int stage) {
        const int bz = b_lo + stage / nk, k0 = (stage % nk) * BK_;
        const float* A = p.A + (long)bz * p.sa;
        const float* B = p.B + (long)bz * p.sb;
        if (avec) {
            static_assert(NA == 4 || NA == 3, "one float4 per thread covers a 64 x 16 tile");
            const int m4 = tid % (BM_ / 4), k = tid / (BM_ / 4);
            float4 v = float4{0.f, 0.f, 0.f, 0.f};
            if (k < BK_ && k0 + k < p.K) v = *reinterpret_cast<const float4*>(A + (long)(k0 + k) * p.ak + m0 + 4 * m4);
            ra[0] = v.x; ra[1] = v.y; ra[2] = v.z; ra[3 % NA] = NA == 4 ? v.w : ra[3 % NA];
        } else if (akvec) {
            const float4 v = *reinterpret_cast<const float4*>(A + (long)(m0 + (tid >> 2)) * p.am + k0 + 4 * (tid & 3));
            ra[0] = v.x; ra[1] = v.y; ra[2] = v.z; ra[3 % NA] = v.w;
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int e = tid + 256 * i;
                int m, k;
                if (p.ak == 1) { k = e & (BK_ - 1); m = e >> 4; } else { m = e % BM_; k = e / BM_; }
                const int gm = m0 + m, gk = k0 + k;
                ra[i] = (gm < p.M && gk < p.K) ? A[gm * p.am + gk * p.ak] : 0.f;
            }
        }
        if (bvec) {
#pragma unroll
            for (int i = 0; i < NB / 4; ++i) {
                const int e = tid + 256 * i;
                const int n4 = e & 31, k = e >> 5;
                float4 v = float4{0.f, 0.f, 0.f, 0.f};
                if (k0 + k < p.K) v = *reinterpret_cast<const float4*>(B + (long)(k0 + k) * p.bk + n0 + 4 * n4);
                rb[4 * i] = v.x; rb[4 * i + 1] = v.y; rb[4 * i + 2] = v.z; rb[4 * i + 3] = v.w;
            }
        } else if (bkvec) {
#pragma unroll
            for (int i = 0; i < NB / 4; ++i) {
                const int e = tid + 256 * i;
                const float4 v = *reinterpret_cast<const float4*>(B + (long)(n0 + (e >> 2)) * p.bn + k0 + 4 * (e & 3));
                rb[4 * i] = v.x; rb[4 * i + 1] = v.y; rb[4 * i + 2] = v.z; rb[4 * i + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int e = tid + 256 * i;
                int n, k;
                if (p.bn == 1) { n = e & (BN_ - 1); k = e >> 7; } else { k = e & (BK_ - 1); n = e >> 4; }
                const int gn = n0 + n, gk = k0 + k;
                rb[i] = (gn < p.N && gk < p.K) ? B[gk * p.bk + gn * p.bn] : 0.f;
            }
        }
    };
    auto commit = [&]() {
        if (avec) {
            const int m4 = tid % (BM_ / 4), k = tid / (BM_ / 4);
            if (k < BK_) *reinterpret_cast<float4*>(&As[k * AP + 4 * m4]) = float4{ra[0], ra[1], ra[2], ra[3 % NA]};
        } else if (akvec) {
#pragma unroll
            for (int j = 0; j < 4; ++j) As[(4 * (tid & 3) + j) * AP + (tid >> 2)] = ra[j % NA];
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int e = tid + 256 * i;
                int m, k;
                if (p.ak == 1) { k = e & (BK_ - 1); m = e >> 4; } else { m = e % BM_; k = e / BM_; }
                As[k * AP + m] = ra[i];
            }
        }
        if (bvec) {
#pragma unroll
            for (int i = 0; i < NB / 4; ++i) {
                const int e = tid + 256 * i;
                const int n4 = e & 31, k = e >> 5;
                *reinterpret_cast<float4*>(&Bs[k * BP + 4 * n4]) = float4{rb[4 * i], rb[4 * i + 1], rb[4 * i + 2], rb[4 * i + 3]};
            }
        } else if (bkvec) {
#pragma unroll
            for (int i = 0; i < NB / 4; ++i) {
                const int e = tid + 256 * i;
#pragma unroll
                for (int j = 0; j < 4; ++j) Bs[(4 * (e & 3) + j) * BP + (e >> 2)] = rb[4 * i + j];
            }
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int e = tid + 256 * i;
                int n, k;
                if (p.bn == 1) { n = e & (BN_ - 1); k = e >> 7; } else { k = e & (BK_ - 1); n = e >> 4; }
                Bs[k * BP + n] = rb[i];
            }
        }
    };
    load(0);
#pragma unroll 1
    for (int stage = 0; stage < nstage; ++stage) {
        __syncthreads();                       // everyone is done reading the previous stage
        commit();
        __syncthreads();
        if (stage + 1 < nstage) load(stage + 1);
#pragma unroll
        for (int ks = 0; ks < BK_; ks += 4) {
            float a[MT], b[2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = As[(ks + g) * AP + mt * 16 + l15];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) b[nt] = Bs[(ks + g) * BP + wave * 32 + nt * 16 + l15];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                if (mt < mt_used) {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
                }
        }
    }
    float* C = p.C + (p.reduce_batch ? 0 : (long)b_lo * p.sc);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gm = m0 + mt * 16 + 4 * g + r;
            if (gm >= p.M) continue;
            float bv = 0.f;
            if (p.bias_mode == 1) bv = p.bias[gm];
            else if (p.bias_mode == 2) bv = p.bias[gm / p.bias_div];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int gn = n0 + wave * 32 + nt * 16 + l15;
                if (gn >= p.N) continue;
                float v = p.alpha * acc[mt][nt][r];
                float* c = C + gm * p.ldc + gn;
                if (p.reduce_batch) {
                    atomicAdd(c, v);
                } else {
                    v += bv;
                    if (p.beta != 0.f) v += p.beta * (*c);
                    if (p.act == TT_ACT_ELU) v = elu1(v);
                    *c = v;
                }
            }
        }
}

}  // namespace

extern "C" int tt_gemm(const float* A, const float* Bm, float* C, const float* bias,
                       int M, int N, int K, int transA, int transB, int64_t lda, int64_t ldb, int64_t ldc,
                       int batch, int64_t sa, int64_t sb, int64_t sc, int reduce_batch,
                       float alpha, float beta, int bias_mode, int bias_div, int act, void* stream) {
    if (!A || !Bm || !C || M <= 0 || N <= 0 || K <= 0 || batch <= 0) return TT_E_BADARG;
    if (bias_mode && !bias) return TT_E_BADARG;
    if (reduce_batch && (bias_mode || act || beta != 1.f)) return TT_E_UNSUPPORTED;   // accumulates into C
    if (batch > 65535) return TT_E_UNSUPPORTED;
    GemmP p{A, Bm, C, bias, M, N, K,
            transA ? 1 : (long)lda, transA ? (long)lda : 1, transB ? 1 : (long)ldb, transB ? (long)ldb : 1, (long)ldc,
            (long)sa, (long)sb, (long)sc, reduce_batch, batch, alpha, beta, bias_mode, bias_div > 0 ? bias_div : 1, act};
    static const bool valu_only = tt_tune_set("TTRAP_GEMM_VALU");      // A/B switch for measurements
    if (!valu_only) {
        // 64 rows per workgroup; row tiles past M are skipped inside the kernel (M = 129: the third block does 1/4 of the MFMAs)
        // reduce_batch: up to 64 batch groups (measured: 64-way atomics per element cost less than longer per-workgroup loops)
        const int bper = reduce_batch ? (batch + 63) / 64 : 1;
        dim3 grid((N + 127) / 128, (M + 63) / 64, (batch + bper - 1) / bper);
        hipLaunchKernelGGL(k_gemm_mfma<64>, grid, dim3(256), 0, tt_stream(stream), p, bper);
        TT_LAUNCH_CHECK();
        return 0;
    }
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, batch);
    hipLaunchKernelGGL(k_gemm, grid, dim3(256), 0, tt_stream(stream), p);
    TT_LAUNCH_CHECK();
    return 0;
}
