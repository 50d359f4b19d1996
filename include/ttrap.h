/*
 * ttrap.h -- C ABI of libttrap_hip.so, the MI355X (gfx950) implementation of the Timbre-Trap
 * hot path:  NSGT constant-Q transform (forward / inverse)  ->  2-D strided-conv autoencoder
 * (forward / backward)  ->  reconstruction / transcription / consistency losses  ->  clip + AdamW.
 *
 * The reference (sony/timbre-trap) is pure Python on stock torch ops and has no FFI of its own;
 * each entry point below names the reference Python it replaces (file:line under /root/reference).
 * The host side that binds these symbols is timbre-trap_amd/timbre_trap/_hip.py (ctypes); the
 * reference-side binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is DEVICE memory owned by the caller (torch tensors), contiguous, fp32 unless
 *     stated; nothing is allocated or retained by the library; kernels are enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the null stream) and the call returns immediately.
 *   - return value: 0 = ok, >0 = hipError_t of the failed launch, <0 = TT_E_* argument error.
 *   - activations are NCHW with W = time contiguous:  (B, C, H, T).
 *   - no CPU fallback exists: without the library the Python layer raises.
 */
#ifndef TTRAP_H
#define TTRAP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TT_E_BADARG      (-1)
#define TT_W_JOIN_LEFT 1       /* tt_wide_level_bwd_gated_join only: the level's backward is done, the riding skip join is NOT (see there) */
#define TT_E_UNSUPPORTED (-2)   /* shape outside the compiled set, or one clip of >= 2^31 elements (32-bit offsets) */

/* flags of tt_resblock_fwd / tt_resblock_bwd: round the operands of the 3x3 convolutions and of dW1 to bf16 for the
 * matrix cores (v_mfma_f32_16x16x32_bf16, fp32 accumulation, fp32 tensors in HBM) at C >= 16.  Default 0 = exact fp32. */
#define TT_FLAG_BF16_OPERANDS 1
/* split-bf16 ("bf16x3"): every fp32 operand is fed to the same instruction as hi + lo (two round-to-nearest bf16) and
 * products are accumulated as hi*hi + lo*hi + hi*lo in fp32 -- fp32-class results (product error ~1e-5 relative,
 * unbiased) at 16/3 of the fp32 matrix rate.  Takes precedence over TT_FLAG_BF16_OPERANDS. */
#define TT_FLAG_BF16_SPLIT    2

#define TT_ACT_NONE 0
#define TT_ACT_ELU  1

/* Library / build identification. */
int         tt_version(void);
const char* tt_arch(void);               /* "gfx950" */
const char* tt_error_string(int code);   /* hipGetErrorString for code > 0 */
/* Test / tuning knob (no reference counterpart): the persistent kernels launch min(work items, CUs * workgroups per CU)
 * workgroups with CUs = 256 on MI355X.  A smaller figure makes every workgroup walk several tiles even on small inputs,
 * which is how the parity tests reach the multi-tile loops (cross-tile LDS-DMA prefetch, XCD-ordered tile walk) at sizes
 * the CPU oracle finishes in seconds.  cus <= 0 only queries.  Returns the previous value.  Process-wide, not stream-ordered. */
int         tt_set_cu_limit(int cus);

/* ------------------------------------------------------------------------------------------------
 * NSGT constant-Q transform.  Replaces cqt_pytorch.CQT.encode / .decode as called from
 * timbre_trap/framework/cqtwrapper.py:67 and :207, fused with CQT.to_real (:74-97) /
 * CQT.to_complex (:99-120) and the inf-norm of CQT.decode (:209-211).
 *
 * Specialised for the reference configuration: block_length N = 66150 (3 s @ 22.05 kHz,
 * N/2 = 33075 = 675 * 49), max_window_length M = 1024.  All conventions of the transform are
 * data (tables built on the host by timbre_trap/framework/nsgt_plan.py):
 *   tw675   [675]   float2  exp(-2 pi i j / 675)
 *   tw49    [49]    float2  exp(-2 pi i j / 49)
 *   twNc    [49][675] float2  exp(-2 pi i n2 k1 / 33075)   (four-step twiddles, row n2, column k1)
 *   twN     [33076] float2  exp(-2 pi i j / 66150)
 *   tw1024  [1024]  float2  exp(-2 pi i j / 1024)
 *   bin_tab [F][4]  int32   {spec_start, pad, length, win_off} per bin
 *   window  [sumL]  float   ragged analysis windows   (forward)
 *   dual    [sumL]  float   ragged synthesis windows  (inverse)
 *   gat_off [33077] int32 , gat_idx [sumL] int32   CSR: spectral index -> ragged positions
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const float*   tw675;
    const float*   tw49;
    const float*   twNc;
    const float*   twN;
    const float*   tw1024;
    const int32_t* bin_tab;
    const float*   window;
    const float*   dual;
    const int32_t* gat_off;
    const int32_t* gat_idx;
    int32_t        n_bins;     /* F */
    int32_t        sum_len;    /* sum of window lengths */
} tt_cqt_plan;

/* bytes of scratch the two calls below need for `n_clips` = B * n_blocks block-clips */
int64_t tt_cqt_scratch_bytes(int n_clips, int n_bins, int sum_len);

/* audio (B, 1, n_blocks*66150) -> out (B, 2, F, n_blocks*1024), ch0 = re, ch1 = im.
 * If out_complex != 0 the output is interleaved complex64 (B, 1, F, n_blocks*1024) instead
 * (CQT.encode, used raw by experiments/sonify.py:94). */
int tt_cqt_forward(const tt_cqt_plan* plan, const float* audio, float* out, void* scratch,
                   int B, int n_blocks, int out_complex, void* stream);

/* coeffs (B, 2, F, n_blocks*1024) [or interleaved complex (B,1,F,T) if in_complex] ->
 * audio (B, 1, n_blocks*66150).  normalize != 0 applies cqtwrapper.py:209-211
 * (divide the whole tensor by its abs-max when that is non-zero) with no host round trip. */
int tt_cqt_inverse(const tt_cqt_plan* plan, const float* coeffs, float* audio, void* scratch,
                   int B, int n_blocks, int in_complex, int normalize, void* stream);

/* The same transform for ANY block length N and frame count M = 2^m (csrc/cqt_generic.hip): the reference constructor takes
 * arbitrary secs_per_block / sample_rate (timbre_trap/framework/cqtwrapper.py:15-48, cqt_pytorch.CQT(block_length=...,
 * power_of_2_length=True) at :31-35); the two entry points above cover N = 66150 / M = 1024 only.  Slow path by design: the
 * length-N DFT as a Bluestein convolution over P = 2^p >= 2N - 1 points, global-memory Stockham passes, one launch per pass.
 * Tables (host-built, timbre_trap/framework/nsgt_plan.py; bin_tab / window / dual / gat_* as above):
 *   chirp   [N]    float2  exp(-i pi n^2 / N)
 *   bfilt   [P]    float2  DFT_P of the wrapped conjugate chirp, divided by P
 *   twP     [P/2]  float2  exp(-2 pi i q / P)          twM [M/2] float2  exp(-2 pi i q / M)
 *   pos_bin [sumL] int32   bin of every ragged window position
 * Same layouts, flags and normalisation rule as tt_cqt_forward / tt_cqt_inverse. */
typedef struct {
    const float*   chirp;
    const float*   bfilt;
    const float*   twP;
    const float*   twM;
    const int32_t* bin_tab;
    const float*   window;
    const float*   dual;
    const int32_t* gat_off;    /* [N/2 + 2] */
    const int32_t* gat_idx;
    const int32_t* pos_bin;
    int32_t        n_bins;
    int32_t        sum_len;
    int32_t        N;
    int32_t        M;
    int32_t        P;
} tt_cqt_gplan;

int64_t tt_cqt_generic_scratch_bytes(const tt_cqt_gplan* plan, int n_clips);
int tt_cqt_generic_forward(const tt_cqt_gplan* plan, const float* audio, float* out, void* scratch,
                           int B, int n_blocks, int out_complex, void* stream);
int tt_cqt_generic_inverse(const tt_cqt_gplan* plan, const float* coeffs, float* audio, void* scratch,
                           int B, int n_blocks, int in_complex, int normalize, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Convolution stack.  Replaces torch.nn.Conv2d / ConvTranspose2d / ELU as used by
 * timbre_trap/framework/modules.py (ResidualConv2dBlock :721-777, EncoderBlock :597-655,
 * DecoderBlock :658-718, Encoder :396-483, Decoder :486-594) and their autograd backward.
 * ---------------------------------------------------------------------------------------------- */

/* One general direct convolution, used for every layer shape of the model and, with swapped
 * weight strides, for every data-gradient:
 *   y[b,co,ho,t] = act( bias[co] + sum_{ci,kh,kw} w[co*ws_co + ci*ws_ci + kh*ws_kh + kw*ws_kw]
 *                                               * x[b,ci,hi,ti] ) + (res ? res[b,co,ho,t] : 0)
 *   ti = t + kw*dil_w - pad_w
 *   transposed == 0 :  hi = ho*stride_h + kh*dil_h - pad_h
 *   transposed == 1 :  hi = (ho + pad_h - kh*dil_h) / stride_h   when divisible (else no term)
 * bias and res may be NULL.  Weight strides are signed (a flipped kernel = negative stride with
 * `w` pointing at the last tap). */
int tt_conv2d(const float* x, const float* w, const float* bias, const float* res, float* y,
              int B, int Cin, int Hin, int T, int Cout, int Hout,
              int KH, int KW, int stride_h, int dil_h, int dil_w, int pad_h, int pad_w,
              int transposed, int64_t ws_co, int64_t ws_ci, int64_t ws_kh, int64_t ws_kw,
              int act, void* stream);

/* Weight + bias gradient of the same general convolution (transposed == 0 form):
 *   dw[co*ws_co + ci*ws_ci + kh*ws_kh + kw*ws_kw] += sum_{b,ho,t} g[b,co,ho,t] * x[b,ci,hi,ti]
 *   dbias[co] += sum g[b,co,:,:]          (dbias may be NULL)
 * Accumulates (+=) into dw/dbias: zero them first for a plain gradient. */
int tt_conv2d_wgrad(const float* x, const float* g, float* dw, float* dbias,
                    int B, int Cin, int Hin, int T, int Cout, int Hout,
                    int KH, int KW, int stride_h, int dil_h, int dil_w, int pad_h, int pad_w,
                    int64_t ws_co, int64_t ws_ci, int64_t ws_kh, int64_t ws_kw, void* stream);

/* g = dy * ELU'(a) expressed through the saved OUTPUT y = ELU(a):  ELU' = y > 0 ? 1 : y + 1. */
int tt_elu_bwd(const float* dy, const float* y, float* g, int64_t n, void* stream);

/* Fused ResidualConv2dBlock forward (modules.py:755-777):
 *   y = ELU(W2 . ELU(W1 (*)_dil x + b1) + b2) + x      x,y: (B,C,H,T), W1 (C,C,3,3), W2 (C,C,1,1)
 * Supported C: 4,8,16,32; dilation 1..3 (other widths: compose tt_conv2d calls).
 * If h1 != NULL the hidden activation ELU(W1 (*) x + b1) (B,C,H,T) is also written, for tt_resblock_bwd. */
int64_t tt_wgrad_scratch_floats(void);

int tt_resblock_fwd(const float* x, const float* w1, const float* b1, const float* w2,
                    const float* b2, float* y, float* h1, int B, int C, int H, int T, int dilation,
                    int flags, void* stream);

/* Fused ResidualConv2dBlock backward.  h1 = the hidden activation saved by tt_resblock_fwd, or NULL to
 * recompute it from x (one more 3x3 convolution, half the saved-activation memory):
 *   inputs  x, h1, dy        outputs  dx (written), dw1/db1/dw2/db2 (accumulated, +=)
 * `ws` is scratch of B*C*H*T + tt_wgrad_scratch_floats() floats (dL/d(conv1 pre-activation), then the
 * per-workgroup partial weight gradients that a second launch sums without atomics). */
int tt_resblock_bwd(const float* x, const float* h1, const float* dy, const float* w1, const float* b1,
                    const float* w2, const float* b2, float* dx, float* dw1, float* db1,
                    float* dw2, float* db2, float* ws, int B, int C, int H, int T, int dilation,
                    int flags, void* stream);

/* bf16-STORAGE residual blocks (C = 4, 8, 16, 32), the "bf16 MFMA conv path" of BASELINE config[2]:
 * activations of a level are bf16, channel-innermost [B][H][T][C] in HBM; weights and their gradients stay fp32 (rounded to
 * bf16 into registers), products accumulate in fp32 (v_mfma_f32_16x16x32_bf16).  Replaces the three ResidualConv2dBlocks of
 * one EncoderBlock / DecoderBlock (modules.py:621-624, 690-693) when ops.WIDE_STORAGE == 'bf16':
 *   tt_wide_pack    x (B,C,H,T) fp32 planar -> out bf16 [B][H][T][C]          tt_wide_unpack   the inverse   (C = 4..64)
 *   tt_wide_rb_fwd  y = ELU(W2 . ELU(W1 (*)_dil x + b1) + b2) + x ; h1 (may be NULL) = ELU(W1 (*) x + b1) saved for backward
 *   tt_wide_rb_bwd  from x, h1, dy: dx (written), dw1 / db1 / dw2 / db2 (fp32, accumulated +=); ws = tt_wide_scratch_bytes
 *                   bytes of scratch (dL/d(conv1 pre-activation) in bf16, then per-workgroup partial gradients). */
int64_t tt_wide_scratch_bytes(int B, int C, int H, int T);
int tt_wide_pack(const float* x, void* out, int B, int C, int H, int T, void* stream);
int tt_wide_unpack(const void* in, float* y, int B, int C, int H, int T, void* stream);
int tt_wide_rb_fwd(const void* x, const float* w1, const float* b1, const float* w2, const float* b2, void* y, void* h1,
                   int B, int C, int H, int T, int dilation, void* stream);
/* The same block with a weighted skip join in its epilogue (round 6): y = block(x) + skip_weights[skip_idx] * skip[b mod skip_B] --
 * the join behind a DecoderBlock (reference modules.py:112 `skip_weights[i] * embedding`, :569-589 `y = y + skip`) without a pass of
 * its own: skip = an encoder embedding of skip_B clips, same C / H / T and element type (B % skip_B == 0; B = 2 skip_B when the decoder
 * runs the reconstruction and the transcription decode as one batch).  skip_weights may be NULL (scale 1).  One rounding of the joined
 * value.  Backward: the block's own backward on the incoming gradient (tt_wide_rb_bwd / tt_wide_level_bwd) + tt_skip_join16_bwd. */
int tt_wide_rb_fwd_join(const void* x, const float* w1, const float* b1, const float* w2, const float* b2, void* y, void* h1,
                        const void* skip, const float* skip_weights, int skip_idx, int skip_B, int B, int C, int H, int T, int dilation,
                        void* stream);
int tt_wide_rb_bwd(const void* x, const void* h1, const void* dy, const float* w1, const float* w2, const float* b2,
                   void* dx, float* dw1, float* db1, float* dw2, float* db2, void* ws, int B, int C, int H, int T,
                   int dilation, void* stream);
/* The backward of ALL residual blocks of one level in one call (modules.py:621-624 / 690-693 differentiated: block nblocks-1 first):
 * tt_wide_rb_bwd for every block -- x[i], h1[i], w1[i], w2[i], b2[i], dilations[i]; dy enters the last block, dx leaves the first,
 * tmp0 / tmp1 (nblocks > 1) are two (B,C,H,T) 16-bit buffers for the gradients between the blocks -- with the partial-sum reduces
 * of all blocks DEFERRED into one launch at the end (two launches fewer per level; same sums in the same order, so every output is
 * bit-identical to nblocks calls of tt_wide_rb_bwd).  ws = tt_wide_level_scratch_bytes(nblocks, B, C, H, T) bytes; nblocks <= 4. */
int64_t tt_wide_level_scratch_bytes(int nblocks, int B, int C, int H, int T);
int tt_wide_level_bwd(int nblocks, const void* const* x, const void* const* h1, const void* dy, const float* const* w1,
                      const float* const* w2, const float* const* b2, void* dx, void* tmp0, void* tmp1, float* const* dw1,
                      float* const* db1, float* const* dw2, float* const* db2, void* ws, int B, int C, int H, int T,
                      const int* dilations, void* stream);
/* The same with the gradient leaving the level ALREADY GATED for the layer in front of it: dx = (d loss / d x[0]) * ELU'(x[0]).  In the
 * reference every residual level but the encoder's first sits behind a layer that ends in an ELU (modules.py:626-630 sconv + ELU ->
 * the next EncoderBlock's block1; :683-693 tconv + ELU -> block1), so x[0] IS that layer's saved output and the product is the first
 * thing its backward computes; done here, in the epilogue of the first block's data gradient where x[0] is at hand, that layer's
 * backward (tt_sconv16_bwd_pregated / tt_tconv16_bwd_pregated below) reads one tensor less and stages nothing through registers.
 * All other outputs as tt_wide_level_bwd. */
int tt_wide_level_bwd_gated(int nblocks, const void* const* x, const void* const* h1, const void* dy, const float* const* w1,
                            const float* const* w2, const float* const* b2, void* dx, void* tmp0, void* tmp1, float* const* dw1,
                            float* const* db1, float* const* dw2, float* const* db2, void* ws, int B, int C, int H, int T,
                            const int* dilations, void* stream);
/* tt_wide_level_bwd_gated with the backward of a weighted skip join riding on it (round 6).  x[0], the level's input, is at the same time
 * the encoder embedding e of a skip connection (reference modules.py:112, :569-589: `y + skip_weights[i] * e` in the decoder); skip_g =
 * the gradient that reached that join's output, skip_reps (1 or 2) batches of B clips back to back (2: the pair decode).  The first block's
 * gated epilogue then writes  dx = (dy + W1^T (*) dA1 + skip_weights[skip_idx] * (g[0] + g[1])) * ELU'(x[0])  -- the embedding's two
 * gradient contributions in one tensor, so that nothing is left to add -- and  skip_dw[skip_idx] += <g[0] + g[1], x[0]>  (times 1 / S under
 * tt_set_loss_scale).  Two more loads per lane and pixel instead of tt_skip_join16_bwd's pass over five tensors.  dilations[0] must be 1
 * (TT_E_UNSUPPORTED before anything is launched otherwise).  Returns TT_W_JOIN_LEFT (> 0) when the level's backward ran but the first
 * block's kernel was one without the riding form (the A/B dispatches TTRAP_DXW=0 / TTRAP_WBWD1 / TTRAP_NARROW_FUSED16=0): the caller then
 * applies the join with tt_skip_join16_bwd(gate | 2) on dx. */
int tt_wide_level_bwd_gated_join(int nblocks, const void* const* x, const void* const* h1, const void* dy, const float* const* w1,
                                 const float* const* w2, const float* const* b2, void* dx, void* tmp0, void* tmp1, float* const* dw1,
                                 float* const* db1, float* const* dw2, float* const* db2, void* ws, int B, int C, int H, int T,
                                 const int* dilations, const void* skip_g, int skip_reps, const float* skip_weights, int skip_idx,
                                 float* skip_dw, void* stream);
/* g *= ELU'(y) in place on n 16-bit elements (n % 8 == 0): the same factor for a gradient that reaches such a layer by another way
 * (a skip connection, a caller's own use of an encoder embedding; ops.GateTapFn). */
int tt_gate16(void* g, const void* y, int64_t n, void* stream);
/* The whole backward of one block in ONE pass from x and dy only (csrc/conv_level_bf16.hip; C = 16, 32, else
 * TT_E_UNSUPPORTED): the hidden activation is recomputed per tile (bit-identical to what tt_wide_rb_fwd would have stored),
 * dL/d(conv1 pre-activation) stays in LDS -- reads x and dy, writes dx.  The forward can then run with h1 = NULL.
 * Same results as tt_wide_rb_bwd (modules.py:755-777 differentiated).  ws = tt_wide_fused_scratch_bytes(C) bytes. */
int64_t tt_wide_fused_scratch_bytes(int C);
int tt_wide_rb_bwd_fused(const void* x, const void* dy, const float* w1, const float* b1, const float* w2, const float* b2,
                         void* dx, float* dw1, float* db1, float* dw2, float* db2, void* ws, int B, int C, int H, int T,
                         int dilation, void* stream);
/* The whole backward of one block in ONE pass from x, the SAVED h1 and dy (csrc/conv_level_bf16.hip k_wrb_bwds; C = 16, 32, else
 * TT_E_UNSUPPORTED): a workgroup walks a strip of 32 columns downwards, the pointwise chain runs once per image row (+ the column
 * halo), dL/d(conv1 pre-activation) lives in a ring of LDS rows only, and the 3x3 weight gradient is indexed by the pixel of x
 * (dW1[tap] = sum_q x[q] (x) dA1[q - tap D]) so that x needs no halo: h1, dy, x in and dx out -- 4 tensors of HBM traffic where
 * tt_wide_rb_bwd's per-stage kernels move 7; nothing is recomputed.  Same results as the per-stage kernels: dx bit-identical, the
 * fp32 weight / bias gradients equal up to summation order (modules.py:755-777 differentiated).
 * ws = tt_wide_onepass_scratch_bytes(C) bytes (<= tt_wide_scratch_bytes).  tt_wide_rb_bwd itself takes this path where
 * tt_wide_rb_bwd_is_onepass(C, dilation) says 1 (measured per width / dilation; TTRAP_WBWD1 = 0 / 1 forces a choice). */
int64_t tt_wide_onepass_scratch_bytes(int C);
int tt_wide_rb_bwd_onepass(const void* x, const void* h1, const void* dy, const float* w1, const float* w2, const float* b2,
                           void* dx, float* dw1, float* db1, float* dw2, float* db2, void* ws, int B, int C, int H, int T,
                           int dilation, void* stream);
int tt_wide_rb_bwd_is_onepass(int C, int dilation);

/* bf16 channel-innermost (4,1) strided / transposed layers between the levels (csrc/conv_stride_bf16.hip): the bf16-storage
 * counterparts of tt_sconv_* / tt_tconv_* below.  x, y, dy, dx are bf16 [B][H][T][channels]; w (2C, C, 4, 1), b and their
 * gradients fp32.  C = the narrower side's channel count (4, 8, 16, 32).
 *   tt_sconv16_fwd  x (B,H,T,C)  -> y (B,(H-4)/2+1,T,2C) = ELU(conv + b)          (modules.py:626-630)
 *   tt_tconv16_fwd  x (B,H,T,2C) -> y (B,2H+2+out_pad,T,C) = ELU(tconv + b)       (modules.py:685-689)
 *   *_bwd           from x, the saved OUTPUT y and dy: dx (written, may be NULL), dw / db (accumulated, +=); the ELU gate
 *                   dy * ELU'(y) is applied on the fly inside the data- and weight-gradient kernels.
 *                   ws: tt_stride16_scratch_bytes(C) bytes (per-wave partial gradients). */
int64_t tt_stride16_scratch_bytes(int C);
int tt_sconv16_fwd(const void* x, const float* w, const float* b, void* y, int B, int C, int H, int T, void* stream);
int tt_sconv16_bwd(const void* x, const void* y, const void* dy, const float* w, void* dx, float* dw, float* db, void* ws,
                   int B, int C, int H, int T, void* stream);
int tt_tconv16_fwd(const void* x, const float* w, const float* b, void* y, int B, int C, int H, int T, int out_pad,
                   void* stream);
int tt_tconv16_bwd(const void* x, const void* y, const void* dy, const float* w, void* dx, float* dw, float* db, void* ws,
                   int B, int C, int H, int T, int out_pad, void* stream);
/* *_bwd_pregated: the same from x and g = dy * ELU'(y) (what tt_wide_level_bwd_gated / tt_latent16_expand_gated leave): the saved
 * output is not needed.  tt_tconv16_bwd_pregated with gate_dx = 1 (C = 16, 32; else TT_E_UNSUPPORTED) passes the favour on: dx leaves
 * as dx * ELU'(x) for the layer in front of the first DecoderBlock (Decoder.convin + ELU, modules.py:534-537 -> tt_latent16_*_pregated). */
int tt_sconv16_bwd_pregated(const void* x, const void* g, const float* w, void* dx, float* dw, float* db, void* ws, int B, int C,
                            int H, int T, void* stream);
int tt_tconv16_bwd_pregated(const void* x, const void* g, const float* w, void* dx, float* dw, float* db, void* ws, int B, int C,
                            int H, int T, int out_pad, int gate_dx, void* stream);

/* The (31,1) latent heads on bf16 channels-last embeddings (csrc/latent_bf16.hip; modules.py:446 Encoder.convlat and :534
 * Decoder.convin).  w is the (D', CT, E, 1) weight of either layer (index (d CT + c) E + h); (CT, D') = (32, <= 48) or (64, <= 144);
 * T % 16 == 0.  ws: tt_latent16_scratch_bytes bytes.
 *   tt_latent16_contract  out (B,Dout,T) fp32 = [bias +] sum_{c,h} w[d][c][h] in[b,h,t,c] for d < Dout (D, or D - 1 to skip the
 *                         gradient of a constant last channel); in = x (cl16), or, with gy != NULL, in = x * ELU'(gy) on the fly
 *                         (data gradient of convin from dy and the saved output)
 *   tt_latent16_expand    out (B,CT,E,T) cl16 = sum_d w[d][c][h] z[b,d,t], with bias != NULL: ELU(bias[c] + .); z (B,Dz,T) with
 *                         Dz = D, or Dz = D - 1 and channel D - 1 = the constant `fill` (the indicator of TimbreTrap.decode,
 *                         modules.py:139-142, without the concatenated tensor)
 *   tt_latent16_wgrad     dw += sum_{b,t} z[b,d,t] g[b,h,t,c]  (z as above; g cl16; with gy != NULL gated and db (CT) += sum g) */
int64_t tt_latent16_scratch_bytes(int B, int CT, int D, int E, int T);
int tt_latent16_contract(const void* in, const void* gy, const float* w, const float* bias, float* out, void* ws, int B, int CT,
                         int D, int Dout, int E, int T, void* stream);
int tt_latent16_expand(const float* z, int Dz, float fill, const float* w, const float* bias, void* out, void* ws, int B, int CT,
                       int D, int E, int T, void* stream);
/* The backward uses with a neighbouring layer's ELU gate moved across the layer boundary (tt_wide_level_bwd_gated for the idea):
 *   tt_latent16_expand_gated      data gradient of Encoder.convlat leaving as (.) * ELU'(gy), gy = the saved output of the strided layer
 *                                 in front (its backward: tt_sconv16_bwd_pregated)
 *   tt_latent16_contract_pregated, tt_latent16_wgrad_pregated   backward of Decoder.convin from g = dy * ELU'(y), as
 *                                 tt_tconv16_bwd_pregated(gate_dx = 1) leaves it; y is not read.  The bias gradient rides as the weight
 *                                 gradient's row of a constant-1 input row: needs D < 48 (CT = 32) / D < 144 (CT = 64), else
 *                                 TT_E_UNSUPPORTED. */
/* 1 where the two pregated entry points below take a head of CT channels and D input channels (they need a free input row of the weight
 * gradient's tile for the bias gradient), else 0 -- what a caller asks before it promises the gate (ops.LatDec16Fn). */
int tt_latent16_pregated_ok(int CT, int D);
int tt_latent16_expand_gated(const float* z, const float* w, const void* gy, void* out, void* ws, int B, int CT, int D, int E, int T,
                             void* stream);
int tt_latent16_contract_pregated(const void* g, const float* w, float* out, void* ws, int B, int CT, int D, int Dout, int E, int T,
                                  void* stream);
int tt_latent16_wgrad_pregated(const float* z, int Dz, float fill, const void* g, float* dw, float* db, void* ws, int B, int CT, int D,
                               int E, int T, void* stream);
int tt_latent16_wgrad(const float* z, int Dz, float fill, const void* g, const void* gy, float* dw, float* db, void* ws, int B,
                      int CT, int D, int E, int T, void* stream);

/* The 3x3 boundary convolutions where fp32 planar tensors meet the bf16 channels-last interior (csrc/conv_edge_bf16.hip), for
 * C0 = 4 first-level channels (model_complexity 2):
 *   tt_convin16_fwd   Encoder.convin (modules.py:433): x (B,2,H,T) fp32 -> y = ELU(conv3x3 + b) as cl16 (B,4,H,T); w (4,2,3,3)
 *   tt_convin16_bwd   from x, the saved output y and dy (cl16): dw, db (+=), dx (B,2,H,T) fp32 (written; may be NULL); y == NULL: dy is
 *                     already dy * ELU'(y), as tt_wide_level_bwd_gated of the first level leaves it -- the saved output is not read
 *   tt_convout16_fwd  Decoder.convout (modules.py:560): x cl16 (B,4,H,T) -> y (B,2,H,T) fp32 = conv3x3 + b; w (2,4,3,3)
 *   tt_convout16_bwd  from x and dy (B,2,H,T) fp32: dx cl16 (written), dw, db (+=)
 * ws: tt_edge16_scratch_bytes() bytes (per-workgroup partial gradients).  fp32 arithmetic; only the cl16 tensors are bf16. */
int64_t tt_edge16_scratch_bytes(void);
int tt_convin16_fwd(const float* x, const float* w, const float* b, void* y, int B, int H, int T, void* stream);
int tt_convin16_bwd(const float* x, const void* y, const void* dy, const float* w, float* dx, float* dw, float* db, void* ws,
                    int B, int H, int T, void* stream);
int tt_convout16_fwd(const void* x, const float* w, const float* b, float* y, int B, int H, int T, void* stream);
int tt_convout16_bwd(const void* x, const float* dy, const float* w, void* dx, float* dw, float* db, void* ws, int B, int H,
                     int T, void* stream);

/* EncoderBlock.sconv (modules.py:626-630): y = ELU(Conv2d(C, 2C, (4,1), stride (2,1))(x) + b).
 * x (B,C,H,T) -> y (B,2C,(H-4)/2+1,T); w (2C,C,4,1).  Supported C: 4,8,16,32. */
int tt_sconv_fwd(const float* x, const float* w, const float* b, float* y, int B, int C, int H, int T,
                 void* stream);
/* Its backward from the saved input x and OUTPUT y: dx (written; may be NULL), dw / db (accumulated, +=);
 * `scratch` holds tt_wgrad_scratch_floats() + numel(dy) floats (reduction partials, then dy * ELU'(y)). */
int tt_sconv_bwd(const float* x, const float* y, const float* dy, const float* w, float* dx, float* dw,
                 float* db, float* scratch, int B, int C, int H, int T, void* stream);

/* DecoderBlock.tconv (modules.py:685-689): y = ELU(ConvTranspose2d(2C, C, (4,1), stride (2,1),
 * output_padding (out_pad,0))(x) + b).  x (B,2C,H,T) -> y (B,C,2H+2+out_pad,T); w (2C,C,4,1). */
int tt_tconv_fwd(const float* x, const float* w, const float* b, float* y, int B, int C, int H, int T,
                 int out_pad, void* stream);
int tt_tconv_bwd(const float* x, const float* y, const float* dy, const float* w, float* dx, float* dw,
                 float* db, float* scratch, int B, int C, int H, int T, int out_pad, void* stream);

/* Batched GEMM for the (31,1) latent layers (modules.py:446 and :534):
 *   for each batch b:  C_b (M,N) = alpha * op(A_b) (M,K) . op(B_b) (K,N) + beta * C_b  [+ bias]
 * row-major, leading dimensions lda/ldb/ldc, batch strides sa/sb/sc (0 = shared operand).
 * If reduce_batch != 0 all batches are summed into ONE C (weight gradients).
 * bias_mode: 0 none, 1 bias[m] per row, 2 bias[m / bias_div] (transposed-conv bias per channel).
 * act = TT_ACT_ELU applies ELU after bias. */
int tt_gemm(const float* A, const float* Bm, float* C, const float* bias,
            int M, int N, int K, int transA, int transB, int64_t lda, int64_t ldb, int64_t ldc,
            int batch, int64_t sa, int64_t sb, int64_t sc, int reduce_batch,
            float alpha, float beta, int bias_mode, int bias_div, int act, void* stream);

/* out[c] += sum over (b, inner) of x[b, c, inner]; x viewed as (B, C, inner). */
int tt_channel_sum(const float* x, float* out, int B, int C, int64_t inner, void* stream);

/* y = a + s[idx] * b  (skip connections, modules.py:112 and :569-589); s may be NULL (scale 1),
 * a may be NULL (treated as 0). */
int tt_scaled_add(const float* a, const float* b, const float* s, int idx, float* y, int64_t n,
                  void* stream);
/* Windowed overlap-add of the half-overlapping chunks of TimbreTrap.chunked_inference (modules.py:259-263):
 *   out[row][i * M/2 + m] += window[m] * chunks[i - c0][row][m]   for chunks i = c0 .. c1-1, accumulated in ascending i
 * chunks (c1-c0, rows, M), window (M), out (rows, n_frames) with rows = B*2*F; M % 8 == 0, n_frames % 4 == 0.  Calls with
 * consecutive chunk ranges reproduce the reference's sequential accumulation bit for bit. */
int tt_window_ola(const float* chunks, const float* window, float* out, int64_t rows, int M, int c0, int c1,
                  int64_t n_frames, void* stream);
/* out[0] += sum(a * b) : gradient of one skip weight. */
int tt_dot(const float* a, const float* b, float* out, int64_t n, void* stream);
/* The same two for the bf16 channels-last path (skip joins under autocast; BASELINE configs[4] trains with
 * skip_connections=True): a, b, y bf16 arrays of n elements (16-byte aligned), s and out fp32; fp32 arithmetic,
 * round-to-nearest-even stores. */
int tt_scaled_add16(const void* a, const void* b, const float* s, int idx, void* y, int64_t n, void* stream);
int tt_dot16(const void* a, const void* b, float* out, int64_t n, void* stream);
/* The weighted skip join as ONE pass each way (round 6; reference modules.py:112 `skip_weights[i] * embedding` and :569-589
 * `y = y + skip`), 16-bit channels-last tensors, n = elements of ONE embedding (n % 8 == 0, 16-byte aligned pointers):
 *   fwd:  out[r*n + i] = y[r*n + i] + s[idx] * e[i]              r < reps (1 or 2)
 *   bwd:  t = sum_r g[r*n + i];   de[i] = s[idx] * t  (* ELU'(e[i]) if gate & 1);   ds[idx] += sum_i t * e[i]      (dy = g: not written)
 *         gate & 2: de[i] += instead of = -- de already holds the other contribution to the embedding's gradient (the data gradient of
 *         the encoder level behind it), so that no separate add is left
 * reps = 2: the decoder runs the reconstruction and the transcription decode of the same latents as one batch of 2 B clips
 * (TimbreTrap.decode_pair) and both halves take the same encoder embedding -- read once, never duplicated.  gate: e is the output of a
 * strided layer + ELU whose backward takes its gradient already multiplied by ELU'(e) (tt_sconv16_bwd_pregated): this contribution
 * then carries the factor as well (what tt_gate16 did in a pass of its own).  s may be NULL (scale 1), de or ds may be NULL.
 * Loss scale (tt_set_loss_scale): g and de are 16-bit gradients and stay scaled, ds *= 1/S. */
int tt_skip_join16_fwd(const void* y, const void* e, const float* s, int idx, void* out, int64_t n, int reps, void* stream);
int tt_skip_join16_bwd(const void* g, const void* e, const float* s, int idx, void* de, float* ds, int64_t n, int reps, int gate,
                       void* stream);

/* ------------------------------------------------------------------------------------------------
 * Objectives.  Replace timbre_trap/framework/objectives.py and TimbreTrap.to_activations
 * (modules.py:271-289).
 * ---------------------------------------------------------------------------------------------- */

/* loss[0] = sum((a-b)^2) * scale     (compute_reconstruction_loss, objectives.py:11-33, with
 * scale = 1/(B*T)).  `partials` is scratch of >= 1024 doubles. */
int tt_sqdiff_sum(const float* a, const float* b, float* loss, double* partials, int64_t n,
                  float scale, void* stream);
/* da = 2*(a-b)*gscale[0]*scale ; db = -da   (either may be NULL) */
int tt_sqdiff_bwd(const float* a, const float* b, const float* gscale, float scale, float* da,
                  float* db, int64_t n, void* stream);
/* Both consistency terms at once (objectives.py:77-104: two squared errors against the same, non-detached, second operand):
 * da1 = 2 scale g1[0] (a1 - b), da2 = 2 scale g2[0] (a2 - b), db = -(da1 + da2); g1 / g2 device scalars (NULL = 0), da1 / da2 / db may be
 * NULL; all pointers 16-byte aligned. */
int tt_sqdiff2_bwd(const float* a1, const float* a2, const float* b, const float* g1, const float* g2, float scale, float* da1,
                   float* da2, float* db, int64_t n, void* stream);
/* The loss(es) AND the gradient(s) in one pass (round 5): the same sums as tt_sqdiff_sum (bit-identical values), and, where a pointer is
 * given, da = 2 scale (a - b), db = -da [tt_sqdiff2_sum_grad: da1, da2 for the two terms against the same b, db = -(da1 + da2);
 * partials: 2048 doubles] -- the gradient for an incoming scalar of 1.  The backward pass is then tt_sqdiff_rescale /
 * tt_sqdiff2_rescale with the incoming scalar(s) on the device: every workgroup reads them and returns when they are 1 (the loss enters
 * the total as it is, train.py:453-466), otherwise the stored gradients are multiplied in place (tt_sqdiff2_rescale recomputes db from
 * da1, da2, which it therefore needs; a NULL scalar counts as 0).  All tensor pointers 16-byte aligned. */
int tt_sqdiff_sum_grad(const float* a, const float* b, float* loss, double* partials, int64_t n, float scale, float* da, float* db,
                       void* stream);
int tt_sqdiff2_sum_grad(const float* a1, const float* a2, const float* b, float* l1, float* l2, double* partials, int64_t n, float scale,
                        float* da1, float* da2, float* db, void* stream);
int tt_sqdiff_rescale(float* da, float* db, const float* g, int64_t n, void* stream);
int tt_sqdiff2_rescale(float* da1, float* da2, float* db, const float* g1, const float* g2, int64_t n, void* stream);

/* act = tanh(sqrt(re^2 + im^2)) for coeffs (B,2,F,T) -> (B,F,T) */
int tt_activations_fwd(const float* coeffs, float* act, int B, int F, int T, void* stream);
int tt_activations_bwd(const float* coeffs, const float* act, const float* dact, float* dcoeffs,
                       int B, int F, int T, void* stream);

/* compute_transcription_loss (objectives.py:36-74). est,tgt: (B,F,T); loss[0] = mean over (B,T)
 * of sum_F w*(est-tgt)^2;  `frame_scale` (B*T floats) receives the per-frame positive scaling
 * for the backward pass; weighted = weight_positive_class. */
int tt_transcription_loss_fwd(const float* est, const float* tgt, float* loss, float* frame_scale,
                              double* partials, int B, int F, int T, int weighted, void* stream);
/* tt_transcription_loss_fwd that also writes dest = the gradient w.r.t. est for an incoming scalar of 1 (round 5; backward is then
 * tt_sqdiff_rescale(dest, NULL, g, B F T): a device-side check of the incoming scalar) */
int tt_transcription_loss_fwd_grad(const float* est, const float* tgt, float* loss, float* frame_scale, double* partials, float* dest,
                                   int B, int F, int T, int weighted, void* stream);
int tt_transcription_loss_bwd(const float* est, const float* tgt, const float* frame_scale,
                              const float* gscale, float* dest, int B, int F, int T, int weighted,
                              void* stream);

/* ------------------------------------------------------------------------------------------------
 * Optimiser.  Replaces torch.nn.utils.clip_grad_norm_(params, max_norm) + torch.optim.AdamW.step
 * (experiments/train.py:334, :493-496) over ONE flat fp32 buffer of n parameters.
 * ---------------------------------------------------------------------------------------------- */
/* norm_out[0] = ||grad||_2 ; partials: >= 1024 doubles of scratch */
int tt_l2norm(const float* x, float* norm_out, double* partials, int64_t n, void* stream);
/* clip coefficient c = min(1, max_norm / (norm[0] + 1e-6)) is applied to grad on the fly (and
 * written back when write_clipped != 0, to match clip_grad_norm_'s in-place semantics). */
int tt_adamw_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, const float* norm,
                  int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                  int step, float max_norm, int write_clipped, int32_t* skipped, void* stream);
/* `skipped` (device, may be NULL): overflow guard of the loss-scaled fp16 backward, the part of torch.amp.GradScaler the reference
 * does without (experiments/train.py:415 has autocast but no scaler).  With norm and skipped given, a NON-FINITE norm[0] makes the
 * call a no-op that adds 1 to skipped[0]; the bias corrections use step - skipped[0], the number of updates actually applied. */

/* Static loss scale of the 16-bit backward, per calling thread (default 1; returns the previous value).  While S != 1 the backward
 * entry points of the 16-bit channels-last layers treat every 16-bit activation gradient they are handed as carrying the factor S
 * and every fp32 gradient as unscaled:
 *   tt_convout16_bwd                      dx (16-bit) = S * (...)            -- the gradient ENTERS the 16-bit region; dw, db unscaled
 *   tt_latent16_expand   (bias == NULL)   out (16-bit) = S * (...)           -- likewise (data gradient of Encoder.convlat)
 *   tt_latent16_wgrad    (gy == NULL)     z (fp32 gradient) is scaled by S on its way to 16 bits; dw *= 1/S
 *   tt_latent16_wgrad    (gy != NULL), tt_latent16_contract (gy != NULL), tt_convin16_bwd, tt_wide_rb_bwd*, tt_wide_level_bwd,
 *   tt_sconv16_bwd, tt_tconv16_bwd, tt_dot16, tt_skip_join16_bwd (ds)
 *                                         every fp32 output (dw, db, dz, dx) *= 1/S; 16-bit data gradients stay scaled
 * With S a power of two this is an exact identity in real arithmetic; in fp16 it lifts activation gradients of ~1e-7 (the loss is a
 * mean over B x T frames) out of the subnormal range.  The value is read on the host when a kernel is launched and passed by value:
 * stream-ordered like any argument, and invisible to other threads.  Forward entry points must be called with S = 1
 * (timbre_trap/framework/ops.py sets it only around its backward calls). */
float tt_set_loss_scale(float scale);

/* ------------------------------------------------------------------------------------------------
 * Helpers on either side of the path.
 * ---------------------------------------------------------------------------------------------- */
/* Per-parameter gradient statistics in ONE launch over the flat gradient buffer (replaces the per-layer
 * `values.grad.norm(2).item()` host syncs of timbre_trap/utils/experiments.py:144-256):
 * for segment s = [offsets[2s], offsets[2s+1]) (element indices into x; 2*n_segments entries, any order, gaps allowed):
 * out[2s] = L2 norm, out[2s+1] = max |g|. */
int tt_segment_stats(const float* x, const int64_t* offsets, int n_segments, float* out, void* stream);
/* Peak picking / thresholding of activations (timbre_trap/utils/processing.py:66-124) on the device.
 * x, out: (n_outer, F, T).  mode 0: keep strict local maxima along F, zero elsewhere (filter_non_peaks);
 * mode 1: x >= threshold (threshold); mode 2: peak && x >= threshold.  Rows f >= f_valid read as zero first (the
 * `activations[valid_freqs] = 0` of experiments/evaluate.py:107-112); f_valid <= 0 or >= F disables the mask. */
int tt_peak_pick(const float* x, float* out, int64_t n_outer, int F, int T, double threshold, int mode, int f_valid,
                 void* stream);
/* Frame-level pitch annotations -> activation targets (timbre_trap/datasets/PitchDataset.py:233-307) in float64:
 * ones at (bins[i], frames[i]), i < n; if radius > 0: correlation along F with the 2*radius+1 `weights` (zero padded,
 * SciPy's symmetric order -> bit-identical to gaussian_filter1d), division by the smallest blurred value over the
 * annotated positions, clip to [0, 1].  out, work: F*T doubles each (work unused when radius == 0). */
int tt_target_activations(const int* bins, const int* frames, int n, const double* weights, int radius, int F, int T,
                          double* work, double* out, void* stream);

/* ---- fp32-class residual blocks on the 16-bit matrix pipe ("x3": split operands), inference ----------------------------------------
 * csrc/conv_x3.hip.  The three ResidualConv2dBlocks of one wide EncoderBlock / DecoderBlock (modules.py:621-624, 690-693; C = 16, 32,
 * dilation 1..3, else TT_E_BADARG / TT_E_UNSUPPORTED) evaluated to fp32 accuracy without the fp32 matrix instructions: every fp32
 * value v (activations in HBM, weights in LDS) is the pair hi = fp16(v), lo = fp16((v - hi) 2^11) and a product is
 * Whi xhi + 2^-11 (Whi xlo + Wlo xhi) on v_mfma_f32_16x16x32_f16 with fp32 accumulators; bias, ELU, residual add in fp32.  Results agree
 * with the fp32 kernels (tt_resblock_fwd) to a few 1e-7 relative; values beyond fp16's range (|v| > 65504) come out non-finite.
 * Forward only (no hidden activation is saved): the no-grad path of ops.residual_level in fp32 mode.
 *   tt_x3_bytes      bytes of one activation tensor in the x3 layout [B][H][T][2][C] halves (= B C H T 4)
 *   tt_x3_pack       x (B,C,H,T) fp32 planar -> x3              tt_x3_unpack   the inverse (exact to 2^-23 relative)
 *   tt_x3_rb_fwd     y = ELU(W2 . ELU(W1 (*)_dil x + b1) + b2) + x; x is an x3 tensor, y an x3 tensor (planar_out = 0; x != y) or an
 *                    fp32 planar (B,C,H,T) tensor (planar_out = 1); weights fp32 (C,C,3,3) / (C,C,1,1)
 *   tt_x3_level_fwd  nblocks blocks (w1[i], b1[i], w2[i], b2[i], dilations[i]); x is fp32 planar (x3_in = 0: packed first) or an x3
 *                    tensor, y fp32 planar (x3_out = 0: written by the last block directly) or an x3 tensor;
 *                    ws = tt_x3_level_scratch_bytes bytes */
int64_t tt_x3_bytes(int B, int C, int H, int T);
int64_t tt_x3_level_scratch_bytes(int B, int C, int H, int T);

/* The NARROW levels (C = 4, 8) of the same no-grad fp32 forward with split operands (csrc/conv_x3.hip, k_x3n_conv: lane = pixel,
 * v_mfma_f32_4x4x4_16b_f16; ResidualConv2dBlock, reference modules.py:721-777, as called from TimbreTrap._inference / evaluate.py:94-95).
 * Between the blocks of a level the activations are "x3n" tensors [B][H][T][2][C] halves (hi plane, lo 2^11 plane); the first block
 * reads the fp32 planar (B,C,H,T) tensor of the layer in front (planar_in), the last one stores fp32 planar (planar_out): no pack /
 * unpack passes.  ws: tt_x3n_level_scratch_bytes.  Same value range as tt_x3_*: |v| <= 65504, non-finite beyond. */
int64_t tt_x3n_level_scratch_bytes(int B, int C, int H, int T);
int tt_x3n_rb_fwd(const void* x, int planar_in, const float* w1, const float* b1, const float* w2, const float* b2, void* y,
                  int planar_out, int B, int C, int H, int T, int dilation, void* stream);
int tt_x3n_level_fwd(int nblocks, const float* x, float* y, const float* const* w1, const float* const* b1, const float* const* w2,
                     const float* const* b2, const int* dilations, void* ws, int B, int C, int H, int T, void* stream);
int tt_x3_pack(const float* x, void* out, int B, int C, int H, int T, void* stream);
int tt_x3_unpack(const void* in, float* y, int B, int C, int H, int T, void* stream);
int tt_x3_rb_fwd(const void* x, const float* w1, const float* b1, const float* w2, const float* b2, void* y, int planar_out, int B,
                 int C, int H, int T, int dilation, void* stream);
int tt_x3_level_fwd(int nblocks, const void* x, int x3_in, void* y, int x3_out, const float* const* w1, const float* const* b1,
                    const float* const* w2, const float* const* b2, const int* dilations, void* ws, int B, int C, int H, int T,
                    void* stream);
/* The strided layers between and above the wide levels with split operands, so that level -> strided layer -> level never leaves the
 * x3 layout.  x is an x3 tensor (planar_in = 0) or -- the layer that ENTERS the split-operand part of the network -- an fp32 planar
 * (B,C,H,T) tensor (planar_in = 1); y an x3 tensor (planar_out = 0) or fp32 planar (planar_out = 1):
 *   tt_x3_sconv_fwd  EncoderBlock.sconv (modules.py:626-630): y = ELU(Conv2d(C, 2C, (4,1), stride (2,1))(x) + bias); x has C channels and
 *                    H >= 4 rows, y 2C channels and (H - 4) / 2 + 1 rows; C = 16, 32 (x3 input) or C = 8 (planar input)
 *   tt_x3_tconv_fwd  DecoderBlock.tconv (modules.py:683-688): y = ELU(ConvTranspose2d(2C, C, (4,1), stride (2,1),
 *                    output_padding (out_pad, 0))(x) + bias); x has 2C channels and H rows, y C channels and 2H + 2 + out_pad rows;
 *                    C = 16 or 32 (x3 input), C = 32 (planar input)
 * other widths: TT_E_UNSUPPORTED.  Weights fp32 in torch's layouts: (2C, C, 4, 1) for both (the transposed layer's is (in, out, 4, 1)). */
int tt_x3_sconv_fwd(const void* x, int planar_in, const float* w, const float* bias, void* y, int planar_out, int B, int C, int H, int T,
                    void* stream);
int tt_x3_tconv_fwd(const void* x, int planar_in, const float* w, const float* bias, void* y, int planar_out, int B, int C, int H, int T,
                    int out_pad, void* stream);

/* The latent heads with split operands (csrc/conv_x3.hip), (C, D) = (64, 128) or (32, 32), else TT_E_UNSUPPORTED / -1:
 *   tt_x3_latent_encode  Encoder.convlat (modules.py:446) = Conv2d(C, D, (E,1)): x an x3 tensor (B, E, T, 2, C), w (D, C, E, 1), bias (D) or
 *                        NULL, z (B, D, T) fp32
 *   tt_x3_latent_decode  Decoder.convin (modules.py:534) = ELU(ConvTranspose2d(D + 1, C, (E,1))): z (B, Dz, T) fp32 with Dz = D + 1, or
 *                        Dz = D and the last input channel constant = fill; w (D + 1, C, E, 1), bias (C) or NULL; y an x3 tensor
 *                        (B, E, T, 2, C) or fp32 planar (B, C, E, T) (planar_out)
 *   ws = tt_x3_latent_scratch_bytes(C, E, D) bytes (the split weights in operand order, rewritten by every call). */
int64_t tt_x3_latent_scratch_bytes(int C, int E, int D);
int tt_x3_latent_encode(const void* x, const float* w, const float* bias, float* z, void* ws, int B, int C, int E, int D, int T,
                        void* stream);
int tt_x3_latent_decode(const float* z, int Dz, float fill, const float* w, const float* bias, void* y, int planar_out, void* ws, int B,
                        int C, int E, int D, int T, void* stream);

/* ---- fp16 twins ---------------------------------------------------------------------------------------------------------------
 * Every entry point of the 16-bit channels-last path above exists a second time with the suffix _h: the same kernels compiled with
 * fp16 elements (csrc/bf16_common.h, -DTT_F16: v_mfma_f32_*_f16, same layouts, same scratch sizes, same argument meaning; `void*`
 * activation pointers then hold IEEE half instead of bf16).  fp16 is the dtype the reference's own train step runs in
 * (experiments/train.py:415: torch.autocast('cuda') defaults to torch.float16): 11 significant bits per stored activation instead of
 * 8, at the price of fp16's range (normal numbers 6.1e-5 .. 65504; the reference uses no GradScaler, and neither does this path).
 * The Python layer picks the set by the autocast dtype. */
int64_t tt_wide_level_scratch_bytes_h(int nblocks, int B, int C, int H, int T);
int tt_wide_level_bwd_h(int nblocks, const void* const* x, const void* const* h1, const void* dy, const float* const* w1,
                        const float* const* w2, const float* const* b2, void* dx, void* tmp0, void* tmp1, float* const* dw1,
                        float* const* db1, float* const* dw2, float* const* db2, void* ws, int B, int C, int H, int T,
                        const int* dilations, void* stream);
int64_t tt_wide_scratch_bytes_h(int B, int C, int H, int T);
int tt_wide_pack_h(const float* x, void* out, int B, int C, int H, int T, void* stream);
int tt_wide_unpack_h(const void* in, float* y, int B, int C, int H, int T, void* stream);
int tt_wide_rb_fwd_h(const void* x, const float* w1, const float* b1, const float* w2, const float* b2, void* y, void* h1,
                   int B, int C, int H, int T, int dilation, void* stream);
int tt_wide_rb_fwd_join_h(const void* x, const float* w1, const float* b1, const float* w2, const float* b2, void* y, void* h1,
                          const void* skip, const float* skip_weights, int skip_idx, int skip_B, int B, int C, int H, int T, int dilation,
                          void* stream);
int tt_wide_rb_bwd_h(const void* x, const void* h1, const void* dy, const float* w1, const float* w2, const float* b2,
                   void* dx, float* dw1, float* db1, float* dw2, float* db2, void* ws, int B, int C, int H, int T,
                   int dilation, void* stream);
int64_t tt_wide_fused_scratch_bytes_h(int C);
int tt_wide_rb_bwd_fused_h(const void* x, const void* dy, const float* w1, const float* b1, const float* w2, const float* b2,
                         void* dx, float* dw1, float* db1, float* dw2, float* db2, void* ws, int B, int C, int H, int T,
                         int dilation, void* stream);
int64_t tt_wide_onepass_scratch_bytes_h(int C);
int tt_wide_rb_bwd_onepass_h(const void* x, const void* h1, const void* dy, const float* w1, const float* w2, const float* b2,
                           void* dx, float* dw1, float* db1, float* dw2, float* db2, void* ws, int B, int C, int H, int T,
                           int dilation, void* stream);
int tt_wide_rb_bwd_is_onepass_h(int C, int dilation);
int64_t tt_stride16_scratch_bytes_h(int C);
int tt_sconv16_fwd_h(const void* x, const float* w, const float* b, void* y, int B, int C, int H, int T, void* stream);
int tt_sconv16_bwd_h(const void* x, const void* y, const void* dy, const float* w, void* dx, float* dw, float* db, void* ws,
                   int B, int C, int H, int T, void* stream);
int tt_tconv16_fwd_h(const void* x, const float* w, const float* b, void* y, int B, int C, int H, int T, int out_pad,
                   void* stream);
int tt_sconv16_bwd_pregated_h(const void* x, const void* g, const float* w, void* dx, float* dw, float* db, void* ws, int B, int C,
                              int H, int T, void* stream);
int tt_gate16_h(void* g, const void* y, int64_t n, void* stream);
int tt_tconv16_bwd_pregated_h(const void* x, const void* g, const float* w, void* dx, float* dw, float* db, void* ws, int B, int C,
                              int H, int T, int out_pad, int gate_dx, void* stream);
int tt_latent16_pregated_ok_h(int CT, int D);
int tt_latent16_expand_gated_h(const float* z, const float* w, const void* gy, void* out, void* ws, int B, int CT, int D, int E, int T,
                               void* stream);
int tt_latent16_contract_pregated_h(const void* g, const float* w, float* out, void* ws, int B, int CT, int D, int Dout, int E, int T,
                                    void* stream);
int tt_latent16_wgrad_pregated_h(const float* z, int Dz, float fill, const void* g, float* dw, float* db, void* ws, int B, int CT, int D,
                                 int E, int T, void* stream);
int tt_wide_level_bwd_gated_h(int nblocks, const void* const* x, const void* const* h1, const void* dy, const float* const* w1,
                              const float* const* w2, const float* const* b2, void* dx, void* tmp0, void* tmp1, float* const* dw1,
                              float* const* db1, float* const* dw2, float* const* db2, void* ws, int B, int C, int H, int T,
                              const int* dilations, void* stream);
int tt_wide_level_bwd_gated_join_h(int nblocks, const void* const* x, const void* const* h1, const void* dy, const float* const* w1,
                                   const float* const* w2, const float* const* b2, void* dx, void* tmp0, void* tmp1, float* const* dw1,
                                   float* const* db1, float* const* dw2, float* const* db2, void* ws, int B, int C, int H, int T,
                                   const int* dilations, const void* skip_g, int skip_reps, const float* skip_weights, int skip_idx,
                                   float* skip_dw, void* stream);
int tt_tconv16_bwd_h(const void* x, const void* y, const void* dy, const float* w, void* dx, float* dw, float* db, void* ws,
                   int B, int C, int H, int T, int out_pad, void* stream);
int64_t tt_latent16_scratch_bytes_h(int B, int CT, int D, int E, int T);
int tt_latent16_contract_h(const void* in, const void* gy, const float* w, const float* bias, float* out, void* ws, int B, int CT,
                         int D, int Dout, int E, int T, void* stream);
int tt_latent16_expand_h(const float* z, int Dz, float fill, const float* w, const float* bias, void* out, void* ws, int B, int CT,
                       int D, int E, int T, void* stream);
int tt_latent16_wgrad_h(const float* z, int Dz, float fill, const void* g, const void* gy, float* dw, float* db, void* ws, int B,
                      int CT, int D, int E, int T, void* stream);
int64_t tt_edge16_scratch_bytes_h(void);
int tt_convin16_fwd_h(const float* x, const float* w, const float* b, void* y, int B, int H, int T, void* stream);
int tt_convin16_bwd_h(const float* x, const void* y, const void* dy, const float* w, float* dx, float* dw, float* db, void* ws,
                    int B, int H, int T, void* stream);
int tt_convout16_fwd_h(const void* x, const float* w, const float* b, float* y, int B, int H, int T, void* stream);
int tt_convout16_bwd_h(const void* x, const float* dy, const float* w, void* dx, float* dw, float* db, void* ws, int B, int H,
                     int T, void* stream);
int tt_scaled_add16_h(const void* a, const void* b, const float* s, int idx, void* y, int64_t n, void* stream);
int tt_dot16_h(const void* a, const void* b, float* out, int64_t n, void* stream);
int tt_skip_join16_fwd_h(const void* y, const void* e, const float* s, int idx, void* out, int64_t n, int reps, void* stream);
int tt_skip_join16_bwd_h(const void* g, const void* e, const float* s, int idx, void* de, float* ds, int64_t n, int reps, int gate,
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TTRAP_H */
