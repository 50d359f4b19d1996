// The (31,1) latent heads on bf16 channels-last embeddings (reference modules.py:446 Encoder.convlat, :534 Decoder.convin).
//
//   convlat   latents[b,d,t] = bias[d] + sum_{c,h} W[d][c][h] top[b,h,t,c]                     top (B,CT,E,T) cl16 -> (B,D,T) fp32
//   convin    y[b,h,t,c]     = ELU(bias[c] + sum_{d} W[d][c][h] z[b,d,t])                      z (B,D+1,T) fp32 -> (B,CT,E,T) cl16
// Both weights are (D', CT, E, 1) arrays with index (d CT + c) E + h, so the six products of the two layers are three kernels:
//   k_lat_contract   out[b,d,t]   = sum_{h,c} W[d][c][h] in[b,h,t,c]    (convlat forward; convin data gradient, gated on the fly)
//   k_lat_expand     out[b,h,t,c] = sum_d W[d][c][h] zt[b,t,d]          (convin forward with bias + ELU; convlat data gradient)
//   k_lat_wgrad      dW[d][c][h] += sum_{b,t} zt[b,t,d] g[b,h,t,c]      (K = pixels, transpose reads; gated g for convin)
// A workgroup owns 256 consecutive frames (a wave 64 = four 16-frame groups) and ALL outputs of them, so the activations are
// read once from HBM; the weights (0.5 MB as bf16, re-laid per call by k_lat_wprep* into the order the lanes consume) stream
// through a double-buffered LDS ring by LDS-DMA and are shared by the four waves.  zt is the small fp32 (B,D',T) operand
// transposed to bf16 [b][t][d] (k_lat_zprep), zero-padded to the K steps and to a pixel stride of 32 KS + 16 elements
// (bank-conflict-free transpose reads in k_lat_wgrad).  fp32 accumulation, bias and ELU; fp32 weight / bias gradients.
#include "bf16_common.h"

namespace {

#ifndef LAT_WGRAD_WAVES
#define LAT_WGRAD_WAVES 4
#endif
#ifndef LAT_EXPAND_Q
#define LAT_EXPAND_Q 2
#endif
#ifndef LAT_NSPLIT
#define LAT_NSPLIT 32
#endif

__device__ __forceinline__ float gate_f(float dy, float y) { return dy * elu_dout(y); }

template <int CT> __device__ __forceinline__ int ochc(int ct, int m) { return (CT / 4) * (m >> 2) + 4 * ct + (m & 3); }

// ---- operand preparation -------------------------------------------------------------------------------------------------
// Wp1[s][dt][lane][8]: step s = h * (CT/32) + half;  value W[d = 16 dt + (lane & 15)][c = 32 half + 8 (lane >> 4) + e][h]
template <int CT, int DT>
__global__ __launch_bounds__(NT) void k_lat_wprep1(const float* __restrict__ w, e16* __restrict__ wp, int D, int E) {
    constexpr int SPH = CT / 32;
    const int i = blockIdx.x * NT + threadIdx.x;                 // one 16-byte piece
    if (i >= E * SPH * DT * 64) return;
    const int lane = i & 63, dt = (i >> 6) % DT, s = (i >> 6) / DT;
    const int h = s / SPH, half = s - h * SPH;
    const int d = 16 * dt + (lane & 15), c0 = 32 * half + 8 * (lane >> 4);
    e16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (e16)(d < D ? w[((long)d * CT + c0 + e) * E + h] : 0.f);
    reinterpret_cast<e16x8*>(wp)[i] = v;
}
// Wp2[h][ct][s][lane][8]: value W[d = 32 s + 8 (lane >> 4) + e][c = ochc(ct, lane & 15)][h]
template <int CT, int KS>
__global__ __launch_bounds__(NT) void k_lat_wprep2(const float* __restrict__ w, e16* __restrict__ wp, int D, int E) {
    constexpr int NC = CT / 16;
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= E * NC * KS * 64) return;
    const int lane = i & 63, s = (i >> 6) % KS, ct = ((i >> 6) / KS) % NC, h = (i >> 6) / (KS * NC);
    const int c = ochc<CT>(ct, lane & 15), d0 = 32 * s + 8 * (lane >> 4);
    e16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (e16)(d0 + e < D ? w[((long)(d0 + e) * CT + c) * E + h] : 0.f);
    reinterpret_cast<e16x8*>(wp)[i] = v;
}
// zt[b][t][ZS] (bf16) = z[b][d][t] (fp32, Dz rows), row Dz = the constant `fill` when Dz < D (the indicator channel of
// TimbreTrap.decode, modules.py:139-142, without materialising the concatenation), zero beyond D
template <int KS>
// `one_row` >= 0: that row (a padding row, >= D) holds the constant 1 -- the weight gradient's row of it is then sum_{b,t} g = the bias
// gradient, which the pregated form has no register staging to sum on the side (tt_latent16_wgrad_pregated)
__global__ __launch_bounds__(NT) void k_lat_zprep(const float* __restrict__ z, e16* __restrict__ zt, int D, int Dz, float fill, int T,
                                                   long npix, float zscale, int one_row = -1) {
    constexpr int ZS = 32 * KS + 16, PCS = ZS / 8;
    // a wave = 64 consecutive frames x one 16-byte piece: the fp32 reads are coalesced, the bf16 writes 16 bytes per lane
    const long i = (long)blockIdx.x * NT + threadIdx.x;
    const long pix = (i / (64 * PCS)) * 64 + (i & 63);
    if (pix >= npix) return;
    const int d0 = (int)((i >> 6) % PCS) * 8;
    const long b = pix / T, t = pix - b * T;
    e16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e)
        v[e] = (e16)(zscale * (d0 + e < Dz ? z[(b * Dz + d0 + e) * T + t] : (d0 + e < D ? fill : (d0 + e == one_row ? 1.f : 0.f))));
    *reinterpret_cast<e16x8*>(zt + pix * ZS + d0) = v;
}

// ---- contract over (h, c) ----------------------------------------------------------------------------------------------------
// QN = 16-frame groups per wave.  With 4 (a workgroup = 256 frames) the 64-clip step launches 256 workgroups = ONE per CU, and the
// 62 barrier-separated K steps run with four waves per CU (452 VGPRs at CT = 64: 164 us for 0.52 GB); QN = 1 gives 1024 workgroups
// of 64 frames, 36 accumulator registers per wave and several workgroups per CU (the weights are re-read from L2).
template <int CT, int DT, bool GATE, int QN>
__global__ __launch_bounds__(NT) void k_lat_contract(const e16* __restrict__ in, const e16* __restrict__ gy,
                                                      const e16* __restrict__ wp, const float* __restrict__ bias,
                                                      float* __restrict__ out, int D, int Dout, int E, int T, long npix, float oscale) {
    constexpr int SPH = CT / 32;
    constexpr int CHUNK = DT * 64 * 16, ROUNDS = (DT * 64 + NT - 1) / NT;      // bytes of one step's weights
    extern __shared__ __align__(16) unsigned char smem[];       // two buffers of ROUNDS * NT * 16 bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int nsteps = E * SPH;
    const long p0 = (long)blockIdx.x * (64 * QN) + wave * (16 * QN);
    long pix[QN]; bool ok[QN]; long base[QN];
#pragma unroll
    for (int q = 0; q < QN; ++q) {
        pix[q] = p0 + 16 * q + n;
        ok[q] = pix[q] < npix;
        const long b = pix[q] / T, t = pix[q] - b * T;
        base[q] = (b * E * T + t) * CT + 8 * g;                  // element offset of row h = 0; + h T CT per row
    }
    auto stage = [&](int s, int buf) {
        const e16* src = wp + (long)s * (DT * 64 * 8);
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int i = r * NT + wave * 64, p = i + lane;
            glds16(src + (long)(p < DT * 64 ? p : DT * 64 - 1) * 8, smem + buf * (ROUNDS * NT * 16) + (long)i * 16);
        }
    };
    auto fetch = [&](int s, e16x8 (&q8)[QN]) {
        const int h = s / SPH, half = s - h * SPH;
#pragma unroll
        for (int q = 0; q < QN; ++q) {
            const long off = base[q] + (long)h * T * CT + 32 * half;
            const bool live = ok[q] && s < nsteps;
            e16x8 v = *reinterpret_cast<const e16x8*>(in + (live ? off : 0));
            if (GATE) {
                const e16x8 yv = *reinterpret_cast<const e16x8*>(gy + (live ? off : 0));
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (e16)gate_f((float)v[e], (float)yv[e]);
            }
            if (!live)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (e16)0.f;
            q8[q] = v;
        }
    };
    f32x4 acc[DT][QN];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int q = 0; q < QN; ++q) acc[dt][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    e16x8 bq[QN], bn[QN];
    stage(0, 0);
    fetch(0, bq);
    for (int s = 0; s < nsteps; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this step's weights (and operands) have landed
        __syncthreads();                                         // ... for every wave; the other buffer is free again
        if (s + 1 < nsteps) stage(s + 1, (s + 1) & 1);
        fetch(s + 1, bn);
        const unsigned char* wb = smem + (s & 1) * (ROUNDS * NT * 16);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const e16x8 a = *reinterpret_cast<const e16x8*>(wb + ((long)dt * 64 + lane) * 16);
#pragma unroll
            for (int q = 0; q < QN; ++q) acc[dt][q] = mma32(a, bq[q], acc[dt][q]);
        }
#pragma unroll
        for (int q = 0; q < QN; ++q) bq[q] = bn[q];
    }
#pragma unroll
    for (int q = 0; q < QN; ++q) {
        if (!ok[q]) continue;
        const long b = pix[q] / T, t = pix[q] - b * T;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = 16 * dt + 4 * g + r;
                if (d < Dout) out[(b * Dout + d) * T + t] = acc[dt][q][r] * oscale + (bias ? bias[d] : 0.f);
            }
    }
    (void)CHUNK;
}

// ---- expand to (h, c) ----------------------------------------------------------------------------------------------------------
// QE = 16-frame groups per wave: 4 (256 frames per workgroup) launches ONE workgroup per CU at the bench shape (81 us for 0.26 GB);
// 2 gives 512 workgroups with half the accumulators.
// GOUT (backward use, !ACT): out = (.) * ELU'(gy) with gy the saved output of the layer in front (Encoder's last strided layer), whose
// backward then takes the gradient already gated (tt_sconv16_bwd_pregated)
template <int CT, int KS, bool ACT, int QE, bool GOUT = false>
__global__ __launch_bounds__(NT) void k_lat_expand(const e16* __restrict__ zt, const e16* __restrict__ wp,
                                                    const float* __restrict__ bias, e16* __restrict__ out, int E, int T,
                                                    long npix, const e16* __restrict__ gy = nullptr) {
    constexpr int NC = CT / 16, NCH = CT / 4, ZS = 32 * KS + 16;
    constexpr int PCS = NC * KS * 64, ROUNDS = (PCS + NT - 1) / NT;            // 16-byte pieces of one row's weights
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const long p0 = (long)blockIdx.x * (64 * QE) + wave * (16 * QE);
    e16x8 bz[QE][KS];
    long obase[QE]; bool ok[QE];
#pragma unroll
    for (int q = 0; q < QE; ++q) {
        const long pix = p0 + 16 * q + n;
        ok[q] = pix < npix;
        const long pc = ok[q] ? pix : 0;
#pragma unroll
        for (int s = 0; s < KS; ++s) bz[q][s] = *reinterpret_cast<const e16x8*>(zt + pc * ZS + 32 * s + 8 * g);
        const long b = pc / T, t = pc - b * T;
        obase[q] = (b * E * T + t) * CT + NCH * g;
    }
    float br[NCH];
#pragma unroll
    for (int e = 0; e < NCH; ++e) br[e] = ACT ? bias[NCH * g + e] : 0.f;
    auto stage = [&](int h, int buf) {
        const e16* src = wp + (long)h * (PCS * 8);
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int i = r * NT + wave * 64, p = i + lane;
            glds16(src + (long)(p < PCS ? p : PCS - 1) * 8, smem + buf * (ROUNDS * NT * 16) + (long)i * 16);
        }
    };
    stage(0, 0);
    for (int h = 0; h < E; ++h) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (h + 1 < E) stage(h + 1, (h + 1) & 1);
        const unsigned char* wb = smem + (h & 1) * (ROUNDS * NT * 16);
        f32x4 acc[NC][QE];
#pragma unroll
        for (int ct = 0; ct < NC; ++ct)
#pragma unroll
            for (int q = 0; q < QE; ++q) acc[ct][q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ct = 0; ct < NC; ++ct)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const e16x8 a = *reinterpret_cast<const e16x8*>(wb + (((long)ct * KS + s) * 64 + lane) * 16);
#pragma unroll
                for (int q = 0; q < QE; ++q) acc[ct][q] = mma32(a, bz[q][s], acc[ct][q]);
            }
#pragma unroll
        for (int q = 0; q < QE; ++q) {
            if (!ok[q]) continue;
            e16* d = out + obase[q] + (long)h * T * CT;
#pragma unroll
            for (int c8 = 0; c8 < NCH / 8; ++c8) {
                e16x8 o, yv;
                if constexpr (GOUT) yv = *reinterpret_cast<const e16x8*>(gy + obase[q] + (long)h * T * CT + 8 * c8);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int ch = 8 * c8 + e;                   // lane channel NCH g + ch = tile ch / 4, row 4g + ch % 4
                    const float a = acc[ch >> 2][q][ch & 3] + br[ch];
                    o[e] = (e16)(ACT ? elu_f(a) : (GOUT ? a * elu_dout((float)yv[e]) : a));
                }
                *reinterpret_cast<e16x8*>(d + 8 * c8) = o;
            }
        }
    }
}

// ---- weight gradient ------------------------------------------------------------------------------------------------------------
template <int CT> __device__ __forceinline__ int gswz(int p) { return CT == 64 ? ((p >> 1) & 3) : ((p >> 2) & 1); }

template <int CT, int DT, int KS, bool GATE>
__global__ __launch_bounds__(NT, LAT_WGRAD_WAVES) void k_lat_wgrad(const e16* __restrict__ zt, const e16* __restrict__ g_in,
                                                   const e16* __restrict__ gy, float* __restrict__ part,
                                                   float* __restrict__ dbpart, int E, int T, long npix, int nsplit) {
    constexpr int NC = CT / 16, ZS = 32 * KS + 16, ZB = ZS * 2, GB = CT * 2;   // bytes per pixel of the two images
    constexpr int ZPC = 64 * ZB / 16, GPC = 64 * GB / 16;                      // pieces per 64-pixel chunk
    constexpr int ZR = (ZPC + NT - 1) / NT, GR = (GPC + NT - 1) / NT;
    constexpr int WPS = 4 / NC;                                  // waves that share a c-tile and split the chunk's 32-pixel halves
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* zs = smem;
    unsigned char* gs = smem + ZR * NT * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4, trj = n >> 2, trq = n & 3;
    const int h = blockIdx.x % E, ps = blockIdx.x / E;
    const int ct = wave % NC, sub0 = wave / NC;
    f32x4 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const e16* zero = reinterpret_cast<const e16*>(&g_wzero16);
    // A chunk is 64 consecutive (clip, frame) pixels: clip and frame of its first pixel are wave-uniform 32-bit scalars, a lane adds its
    // pixel's offset and wraps once (T >= 64) -- the per-piece 64-bit division pixel / T this replaces was most of the staging's
    // ~400 vector instructions per chunk, next to 18 matrix instructions of work (npix < 2^31: checked by the launcher).
    const int npx = (int)npix, nchunks = (npx + 63) / 64;
    for (int ck = ps; ck < nchunks; ck += nsplit) {
        const int px0 = ck * 64;
        const int b0 = px0 / T, t00 = px0 - b0 * T;
        __syncthreads();
        const bool full = px0 + 64 <= npx;                        // every chunk but possibly the last
        const e16* zc = zt + (long)px0 * ZS;
#pragma unroll
        for (int r = 0; r < ZR; ++r) {                           // zt rows of the chunk are one contiguous run
            const int i = r * NT + wave * 64, p = i + lane;
            const bool okp = p < ZPC && (full || px0 + p * 16 / ZB < npx);
            glds16(okp ? zc + p * 8 : zero, zs + (long)i * 16);
        }
        e16x8 v[GATE ? GR : 1], yv[GATE ? GR : 1];
#pragma unroll
        for (int r = 0; r < GR; ++r) {
            const int i = r * NT + wave * 64, p = i + lane;
            const int q = p / (GB / 16), cgp = p % (GB / 16);
            const int cg = (((cgp >> 1) ^ gswz<CT>(q)) << 1) | (cgp & 1);
            const bool okp = p < GPC && (full || px0 + q < npx);
            int b = b0, t = t00 + q;
            if (T >= 64) { if (t >= T) { t -= T; ++b; } }
            else { const int e = t / T; b += e; t -= e * T; }
            const long off = okp ? (long)((b * E + h) * T + t) * CT + cg * 8 : 0;
            if constexpr (!GATE) glds16(okp ? g_in + off : zero, gs + (long)i * 16);
            else { v[r] = *reinterpret_cast<const e16x8*>(g_in + off); yv[r] = *reinterpret_cast<const e16x8*>(gy + off); }
        }
        if constexpr (GATE) {
#pragma unroll
            for (int r = 0; r < GR; ++r) {
                const int p = r * NT + tid;
                const bool okp = p < GPC && px0 + p / (GB / 16) < npx;
                e16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float gq = okp ? gate_f((float)v[r][e], (float)yv[r][e]) : 0.f;
                    dbacc[e] += gq;
                    o[e] = (e16)gq;
                }
                if (p < GPC) *reinterpret_cast<e16x8*>(gs + (long)p * 16) = o;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int sub = sub0; sub < 2; sub += WPS) {              // 32-pixel halves of the chunk
            s16x4 lo, hi;
            {
                const int p = 32 * sub + 4 * g + trj;
                lo = lds_tr16(gs + (long)p * GB + ((ct ^ gswz<CT>(p)) << 5) + 8 * trq);
                hi = lds_tr16(gs + (long)(p + 16) * GB + ((ct ^ gswz<CT>(p + 16)) << 5) + 8 * trq);
            }
            const e16x8 bq = __builtin_bit_cast(e16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int p = 32 * sub + 4 * g + trj;
                const s16x4 alo = lds_tr16(zs + (long)p * ZB + dt * 32 + 8 * trq);
                const s16x4 ahi = lds_tr16(zs + (long)(p + 16) * ZB + dt * 32 + 8 * trq);
                acc[dt] = mma32(__builtin_bit_cast(e16x8, __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7)), bq, acc[dt]);
            }
        }
    }
    float* pw = part + ((long)blockIdx.x * 4 + wave) * (DT * 256);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) pw[(dt * 4 + r) * 64 + lane] = acc[dt][r];
    if constexpr (GATE) {
        // a thread always stages the same physical 16-byte position (NT pieces per round = a multiple of 8 pixels)
        __syncthreads();
        float* dl = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int e = 0; e < 8; ++e) dl[tid * 8 + e] = dbacc[e];
        __syncthreads();
        if (tid < CT) {
            float sum = 0.f;
            for (int t2 = 0; t2 < NT; ++t2) {
                const int q = t2 / (GB / 16), cgp = t2 % (GB / 16);
                const int cg = (((cgp >> 1) ^ gswz<CT>(q)) << 1) | (cgp & 1);
                if (cg == (tid >> 3)) sum += dl[t2 * 8 + (tid & 7)];
            }
            dbpart[(long)blockIdx.x * 64 + tid] = sum;
        }
    }
}

template <int CT, int DT>
__global__ __launch_bounds__(NT) void k_lat_wred(const float* __restrict__ part, const float* __restrict__ dbpart, float* __restrict__ dw,
                                                  float* __restrict__ db, int D, int E, int nsplit, float scale, int db_row = -1,
                                                  float* __restrict__ dbrows = nullptr) {
    constexpr int NC = CT / 16, WPS = 4 / NC, WD = DT * 256;
    const int i = blockIdx.x * NT + threadIdx.x;                 // (h, ct, dump element)
    const int total = E * NC * WD;
    if (i < total) {
        const int e = i % WD, ct = (i / WD) % NC, h = i / (WD * NC);
        float sum = 0.f;
        for (int ps = 0; ps < nsplit; ++ps)
#pragma unroll
            for (int w2 = 0; w2 < WPS; ++w2) sum += part[(((long)ps * E + h) * 4 + ct + NC * w2) * WD + e];
        const int lane = e & 63, r = (e >> 6) & 3, dt = e >> 8;
        const int d = 16 * dt + 4 * (lane >> 4) + r, c = 16 * ct + (lane & 15);
        if (d < D) dw[((long)d * CT + c) * E + h] += sum * scale;
        else if (d == db_row) dbrows[h * CT + c] = sum * scale;   // the row of ones (k_lat_zprep): the sum of g over row h of the embedding; k_lat_dbrows
                                                                   // adds the E rows of a channel in a fixed order (round 5: one atomicAdd per row -- E
                                                                   // contributions per channel in arrival order, low bits varying from run to run; advisor)
    } else if (db && db_row < 0 && i < total + CT * 16) {       // sixteen slices of the workgroups per channel, joined by atomics
        const int c = (i - total) % CT, sl = (i - total) / CT;
        float sum = 0.f;
        for (int wg = sl; wg < nsplit * E; wg += 16) sum += dbpart[(long)wg * 64 + c];
        atomicAdd(db + c, sum * scale);
    }
}

// db[c] += sum_h rows[h][c], one thread per channel, ascending h (the pregated bias gradient's second stage)
__global__ __launch_bounds__(64) void k_lat_dbrows(const float* __restrict__ rows, float* __restrict__ db, int E, int CT) {
    const int c = threadIdx.x;
    if (c >= CT) return;
    float t = 0.f;
    for (int h = 0; h < E; ++h) t += rows[h * CT + c];
    db[c] += t;
}

// ---- launchers ---------------------------------------------------------------------------------------------------------------
constexpr int NSPLIT = LAT_NSPLIT;

template <int CT, int DT, int KS> struct LatSizes {
    static constexpr int SPH = CT / 32, NC = CT / 16, ZS = 32 * KS + 16;
    static long wp1_bytes(int E) { return (long)E * SPH * DT * 64 * 16; }
    static long wp2_bytes(int E) { return (long)E * NC * KS * 64 * 16; }
    static long zt_bytes(long npix) { return npix * ZS * 2 + 4096; }
    static long part_bytes(int E) { return ((long)NSPLIT * E * 4 * DT * 256 + (long)NSPLIT * E * 64) * 4; }
};

template <int CT, int DT, int KS, bool GATE>
int run_contract(const e16* in, const e16* gy, const float* w, const float* bias, float* out, unsigned char* ws, int B, int D,
                 int Dout, int E, int T, hipStream_t st, bool pregated = false) {
    using L = LatSizes<CT, DT, KS>;
    const long npix = (long)B * T;
    e16* wp = reinterpret_cast<e16*>(ws);
    const int pieces = E * L::SPH * DT * 64;
    hipLaunchKernelGGL((k_lat_wprep1<CT, DT>), dim3((pieces + NT - 1) / NT), dim3(NT), 0, st, w, wp, D, E);
    TT_LAUNCH_CHECK();
    constexpr int ROUNDS = (DT * 64 + NT - 1) / NT, LDS = 2 * ROUNDS * NT * 16;
    static AttrOnce once;
#ifndef TT_LAT_QN64
#define TT_LAT_QN64 1
#endif
    constexpr int QN = CT == 64 ? TT_LAT_QN64 : 4;
    auto kern = k_lat_contract<CT, DT, GATE, QN>;
    if (int rc = raise_lds(kern, LDS, once)) return rc;
    // GATE = the backward use (dz of Decoder.convin from the S-scaled gradient of its output): the fp32 result leaves the scaled region
    const float oscale = (GATE || pregated) ? tt_loss_unscale() : 1.f;
    hipLaunchKernelGGL(kern, dim3((unsigned)((npix + 64 * QN - 1) / (64 * QN))), dim3(NT), LDS, st, in, gy, wp, bias, out, D, Dout, E, T, npix, oscale);
    TT_LAUNCH_CHECK();
    return 0;
}

template <int CT, int DT, int KS, bool ACT, bool GOUT = false>
int run_expand(const float* z, int Dz, float fill, const float* w, const float* bias, e16* out, unsigned char* ws, int B, int D, int E,
               int T, hipStream_t st, const e16* gy = nullptr) {
    using L = LatSizes<CT, DT, KS>;
    const long npix = (long)B * T;
    e16* wp = reinterpret_cast<e16*>(ws);
    e16* zt = reinterpret_cast<e16*>(ws + ((L::wp2_bytes(E) + 255) / 256) * 256);
    const int pieces = E * L::NC * KS * 64;
    hipLaunchKernelGGL((k_lat_wprep2<CT, KS>), dim3((pieces + NT - 1) / NT), dim3(NT), 0, st, w, wp, D, E);
    TT_LAUNCH_CHECK();
    const long zp = ((npix + 63) / 64) * 64 * (L::ZS / 8);
    // !ACT = the backward use (dx of Encoder.convlat from the fp32 gradient of the latents): the 16-bit result ENTERS the scaled region
    hipLaunchKernelGGL((k_lat_zprep<KS>), dim3((unsigned)((zp + NT - 1) / NT)), dim3(NT), 0, st, z, zt, D, Dz, fill, T, npix,
                       ACT ? 1.f : tt_loss_scale());
    TT_LAUNCH_CHECK();
    constexpr int PCS = L::NC * KS * 64, ROUNDS = (PCS + NT - 1) / NT, LDS = 2 * ROUNDS * NT * 16;
    static AttrOnce once;
    constexpr int QE = LAT_EXPAND_Q;
    auto kern = k_lat_expand<CT, KS, ACT, QE, GOUT>;
    if (int rc = raise_lds(kern, LDS, once)) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)((npix + 64 * QE - 1) / (64 * QE))), dim3(NT), LDS, st, zt, wp, bias, out, E, T, npix, gy);
    TT_LAUNCH_CHECK();
    return 0;
}

template <int CT, int DT, int KS, bool GATE>
int run_wgrad(const float* z, int Dz, float fill, const e16* g_in, const e16* gy, float* dw, float* db, unsigned char* ws, int B,
              int D, int E, int T, hipStream_t st, bool pregated = false) {
    using L = LatSizes<CT, DT, KS>;
    const long npix = (long)B * T;
    if (npix >= (1l << 30) || (long)B * E * T >= (1l << 31)) return TT_E_UNSUPPORTED;      // 32-bit pixel arithmetic in k_lat_wgrad
    e16* zt = reinterpret_cast<e16*>(ws);
    float* part = reinterpret_cast<float*>(ws + ((L::zt_bytes(npix) + 255) / 256) * 256);
    float* dbpart = part + (long)NSPLIT * E * 4 * DT * 256;
    const long zp = ((npix + 63) / 64) * 64 * (L::ZS / 8);
    // GATE: z holds activations (the latents) and g the S-scaled gradient; !GATE (Encoder.convlat): z IS the gradient, fp32 and unscaled --
    // it takes the factor S on its way to 16 bits.  Either way the dumps carry exactly one S, removed by the reduce.
    // pregated (!GATE kernels, the gated form's meaning): z holds activations, g arrives gated and S-scaled; the bias gradient is the weight
    // gradient's row of a constant-1 input row (needs a padding row: D < 16 DT)
    if (pregated && (GATE || D >= 16 * DT || !db)) return TT_E_UNSUPPORTED;
    const int one_row = pregated ? D : -1;
    hipLaunchKernelGGL((k_lat_zprep<KS>), dim3((unsigned)((zp + NT - 1) / NT)), dim3(NT), 0, st, z, zt, D, Dz, fill, T, npix,
                       (GATE || pregated) ? 1.f : tt_loss_scale(), one_row);
    TT_LAUNCH_CHECK();
    constexpr int ZB = L::ZS * 2, GBY = CT * 2;
    constexpr int ZR = (64 * ZB / 16 + NT - 1) / NT, GR = (64 * GBY / 16 + NT - 1) / NT;
    constexpr int LDS0 = (ZR + GR) * NT * 16, LDS = LDS0 > NT * 32 ? LDS0 : NT * 32;
    static AttrOnce once;
    auto kern = k_lat_wgrad<CT, DT, KS, GATE>;
    if (int rc = raise_lds(kern, LDS, once)) return rc;
    hipLaunchKernelGGL(kern, dim3(E * NSPLIT), dim3(NT), LDS, st, zt, g_in, gy, part, dbpart, E, T, npix, NSPLIT);
    TT_LAUNCH_CHECK();
    const int total = E * (CT / 16) * DT * 256 + CT * 16;
    // (pregated: dbpart, unused by that form's weight-gradient kernel, holds the E x CT row sums between the two stages)
    hipLaunchKernelGGL((k_lat_wred<CT, DT>), dim3((total + NT - 1) / NT), dim3(NT), 0, st, part, dbpart, dw, (GATE || pregated) ? db : nullptr, D, E,
                       NSPLIT, tt_loss_unscale(), one_row, dbpart);
    TT_LAUNCH_CHECK();
    if (pregated) {
        static_assert(CT <= 64, "one wave sums the channels");
        hipLaunchKernelGGL(k_lat_dbrows, dim3(1), dim3(64), 0, st, dbpart, db, E, CT);
        TT_LAUNCH_CHECK();
    }
    return 0;
}

inline int cfg_of(int CT, int D) {                               // 1: CT 32, D <= 48;  2: CT 64, D <= 144;  0: unsupported
    if (CT == 32 && D >= 1 && D <= 48) return 1;
    if (CT == 64 && D >= 1 && D <= 144) return 2;
    return 0;
}

}  // namespace

extern "C" {

int64_t tt_latent16_scratch_bytes(int B, int CT, int D, int E, int T) {
    const long npix = (long)B * T;
    switch (cfg_of(CT, D)) {
        case 1: { using L = LatSizes<32, 3, 2>; return L::wp1_bytes(E) + L::wp2_bytes(E) + L::zt_bytes(npix) + L::part_bytes(E) + 4096; }
        case 2: { using L = LatSizes<64, 9, 5>; return L::wp1_bytes(E) + L::wp2_bytes(E) + L::zt_bytes(npix) + L::part_bytes(E) + 4096; }
    }
    return -1;
}

/* out (B,Dout,T) fp32 = [bias +] contraction of in (B,CT,E,T) cl16 [gated by the saved output gy when gy != NULL] with the first Dout
 * rows of w (D,CT,E,1); Dout = D, or D - 1 to skip the gradient of a constant last input channel */
int tt_latent16_contract(const void* in, const void* gy, const float* w, const float* bias, float* out, void* ws, int B, int CT, int D,
                         int Dout, int E, int T, void* stream) {
    if (!in || !w || !out || !ws || B <= 0 || E <= 0 || T <= 0 || T % 16 || Dout < 1 || Dout > D) return TT_E_BADARG;
    hipStream_t st = tt_stream(stream);
    const e16 *i = (const e16*)in, *y = (const e16*)gy;
    unsigned char* s = (unsigned char*)ws;
    switch (cfg_of(CT, D)) {
        case 1: return y ? run_contract<32, 3, 2, true>(i, y, w, bias, out, s, B, D, Dout, E, T, st) : run_contract<32, 3, 2, false>(i, y, w, bias, out, s, B, D, Dout, E, T, st);
        case 2: return y ? run_contract<64, 9, 5, true>(i, y, w, bias, out, s, B, D, Dout, E, T, st) : run_contract<64, 9, 5, false>(i, y, w, bias, out, s, B, D, Dout, E, T, st);
    }
    return TT_E_UNSUPPORTED;
}

/* out (B,CT,E,T) cl16 = [ELU(bias + .) if bias != NULL] expansion of z with w (D,CT,E,1); z is (B,Dz,T) fp32 with Dz = D, or
 * Dz = D - 1 and the last input channel is the constant `fill` */
int tt_latent16_expand(const float* z, int Dz, float fill, const float* w, const float* bias, void* out, void* ws, int B, int CT, int D,
                       int E, int T, void* stream) {
    if (!z || !w || !out || !ws || B <= 0 || E <= 0 || T <= 0 || T % 16 || (Dz != D && Dz != D - 1)) return TT_E_BADARG;
    hipStream_t st = tt_stream(stream);
    e16* o = (e16*)out;
    unsigned char* s = (unsigned char*)ws;
    switch (cfg_of(CT, D)) {
        case 1: return bias ? run_expand<32, 3, 2, true>(z, Dz, fill, w, bias, o, s, B, D, E, T, st) : run_expand<32, 3, 2, false>(z, Dz, fill, w, bias, o, s, B, D, E, T, st);
        case 2: return bias ? run_expand<64, 9, 5, true>(z, Dz, fill, w, bias, o, s, B, D, E, T, st) : run_expand<64, 9, 5, false>(z, Dz, fill, w, bias, o, s, B, D, E, T, st);
    }
    return TT_E_UNSUPPORTED;
}

/* dw (D,CT,E,1) += sum_{b,t} z x g (B,CT,E,T) cl16 [gated by gy when gy != NULL, then also db (CT) += sum g]; z as in tt_latent16_expand */
int tt_latent16_wgrad(const float* z, int Dz, float fill, const void* g, const void* gy, float* dw, float* db, void* ws, int B, int CT,
                      int D, int E, int T, void* stream) {
    if (!z || !g || !dw || !ws || (gy && !db) || B <= 0 || E <= 0 || T <= 0 || T % 16 || (Dz != D && Dz != D - 1)) return TT_E_BADARG;
    hipStream_t st = tt_stream(stream);
    const e16 *gi = (const e16*)g, *y = (const e16*)gy;
    unsigned char* s = (unsigned char*)ws;
    switch (cfg_of(CT, D)) {
        case 1: return y ? run_wgrad<32, 3, 2, true>(z, Dz, fill, gi, y, dw, db, s, B, D, E, T, st) : run_wgrad<32, 3, 2, false>(z, Dz, fill, gi, y, dw, db, s, B, D, E, T, st);
        case 2: return y ? run_wgrad<64, 9, 5, true>(z, Dz, fill, gi, y, dw, db, s, B, D, E, T, st) : run_wgrad<64, 9, 5, false>(z, Dz, fill, gi, y, dw, db, s, B, D, E, T, st);
    }
    return TT_E_UNSUPPORTED;
}

/* The three backward uses with the ELU gate of a neighbouring layer moved (ops.GateLink; tt_wide_level_bwd_gated for the idea):
 *   tt_latent16_expand_gated     data gradient of Encoder.convlat, leaving as (.) * ELU'(gy): gy = the saved output of the strided layer
 *                                in front, whose backward is then tt_sconv16_bwd_pregated
 *   tt_latent16_contract_pregated / tt_latent16_wgrad_pregated   backward of Decoder.convin from g = dy * ELU'(y) as the transposed layer
 *                                behind it leaves it (tt_tconv16_bwd_pregated, gate_dx = 1): y is not read; needs D < 16 * ceil(D / 16)
 *                                rounded up to the kernel's tile count (a free input row carries the bias gradient): TT_E_UNSUPPORTED else */
/* 1 if tt_latent16_contract_pregated / tt_latent16_wgrad_pregated take a head of CT channels and D input channels (a free input row
 * must be left for the bias gradient: D < 16 * the kernel's row-tile count), else 0 */
int tt_latent16_pregated_ok(int CT, int D) {
    switch (cfg_of(CT, D)) {
        case 1: return D < 16 * 3;
        case 2: return D < 16 * 9;
    }
    return 0;
}

int tt_latent16_expand_gated(const float* z, const float* w, const void* gy, void* out, void* ws, int B, int CT, int D, int E, int T,
                             void* stream) {
    if (!z || !w || !gy || !out || !ws || B <= 0 || E <= 0 || T <= 0 || T % 16) return TT_E_BADARG;
    hipStream_t st = tt_stream(stream);
    switch (cfg_of(CT, D)) {
        case 1: return run_expand<32, 3, 2, false, true>(z, D, 0.f, w, nullptr, (e16*)out, (unsigned char*)ws, B, D, E, T, st, (const e16*)gy);
        case 2: return run_expand<64, 9, 5, false, true>(z, D, 0.f, w, nullptr, (e16*)out, (unsigned char*)ws, B, D, E, T, st, (const e16*)gy);
    }
    return TT_E_UNSUPPORTED;
}

int tt_latent16_contract_pregated(const void* g, const float* w, float* out, void* ws, int B, int CT, int D, int Dout, int E, int T,
                                  void* stream) {
    if (!g || !w || !out || !ws || B <= 0 || E <= 0 || T <= 0 || T % 16 || Dout < 1 || Dout > D) return TT_E_BADARG;
    hipStream_t st = tt_stream(stream);
    switch (cfg_of(CT, D)) {
        case 1: return run_contract<32, 3, 2, false>((const e16*)g, nullptr, w, nullptr, out, (unsigned char*)ws, B, D, Dout, E, T, st, true);
        case 2: return run_contract<64, 9, 5, false>((const e16*)g, nullptr, w, nullptr, out, (unsigned char*)ws, B, D, Dout, E, T, st, true);
    }
    return TT_E_UNSUPPORTED;
}

int tt_latent16_wgrad_pregated(const float* z, int Dz, float fill, const void* g, float* dw, float* db, void* ws, int B, int CT, int D,
                               int E, int T, void* stream) {
    if (!z || !g || !dw || !db || !ws || B <= 0 || E <= 0 || T <= 0 || T % 16 || (Dz != D && Dz != D - 1)) return TT_E_BADARG;
    hipStream_t st = tt_stream(stream);
    switch (cfg_of(CT, D)) {
        case 1: return run_wgrad<32, 3, 2, false>(z, Dz, fill, (const e16*)g, nullptr, dw, db, (unsigned char*)ws, B, D, E, T, st, true);
        case 2: return run_wgrad<64, 9, 5, false>(z, Dz, fill, (const e16*)g, nullptr, dw, db, (unsigned char*)ws, B, D, E, T, st, true);
    }
    return TT_E_UNSUPPORTED;
}

}  // extern "C"
