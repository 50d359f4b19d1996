"""
torch.autograd bindings of the gfx950 kernels (C ABI: include/ttrap.h).

Every Function below owns a forward and a hand-written backward that call straight into
libttrap_hip.so on the current HIP stream; torch only allocates the tensors.  Nothing here has a
CPU path -- tensors must live on the GPU.
"""

import ctypes
import math
import os
import threading
from dataclasses import dataclass

import torch

from .. import _hip
from .._hip import check, ptr, stream_ptr

ACT_NONE, ACT_ELU = 0, 1

# The fused ResidualConv2dBlock kernels (csrc/conv_mfma.hip, csrc/conv_small.hip) are the default; TTRAP_FUSED=0 composes
# the block from the general convolution kernels instead (used to cross-check the two on the GPU).
FUSED_RESBLOCK = os.environ.get('TTRAP_FUSED', '1') != '0'
FUSED_CHANNELS = (4, 8, 16, 32)
# Keep the hidden activation of every residual block for backward (one more (B,C,H,T) tensor per block, no 3x3
# recompute).  TTRAP_SAVE_HIDDEN=0 recomputes instead and halves the residual-block activation memory.
SAVE_HIDDEN = os.environ.get('TTRAP_SAVE_HIDDEN', '1') != '0'
# Arithmetic of the wide (C >= 16) residual blocks on the matrix cores:
#   'fp32'    v_mfma_f32_16x16x4_f32, exact fp32 -- what every parity test pins
#   'bf16x3'  fp32 tensors, every operand fed as hi + lo bf16 (three products): fp32-class results at 16/3 of the fp32 rate
#   'bf16'    operands rounded to bf16, fp32 accumulation (v_mfma_f32_16x16x32_bf16)
#   'fp16'    the same kernels with fp16 elements (v_mfma_f32_16x16x32_f16; the _h entry points of include/ttrap.h)
#   'auto'    (default) inside a ``torch.autocast('cuda')`` region the region's own dtype -- 'fp16' for the reference's unmodified
#             ``torch.autocast('cuda')`` (experiments/train.py:415 takes torch's default, float16, no GradScaler), 'bf16' for
#             ``torch.autocast('cuda', dtype=torch.bfloat16)`` (bench.py, BASELINE config[2]) -- and 'fp32' outside: the reference
#             runs its train step under autocast and everything else in fp32 (experiments/evaluate.py), so the unmodified scripts
#             get the same split here.
PRECISION = os.environ.get('TTRAP_PRECISION', 'auto')
# Storage of the activations INSIDE the wide levels (C = 16, 32: the three residual blocks of an Encoder/DecoderBlock):
#   'fp32'  channel-planar fp32 tensors, one ResBlockFn per block (any precision above)
#   'bf16'  bf16 channel-innermost tensors in HBM, csrc/conv_wide_bf16.hip: the "bf16 MFMA conv path" of BASELINE config[2].
#   'fp16'  the same with fp16 elements.
# Default: follows the precision ('bf16' / 'fp16' mean 16-bit operands AND 16-bit storage); TTRAP_WIDE_STORAGE overrides.
WIDE_STORAGE = os.environ.get('TTRAP_WIDE_STORAGE', '')
WIDE_CHANNELS = (4, 8, 16, 32)        # every level of the model (C = 4 needs an even number of frames)


def precision():
    """The arithmetic in force for the call being made: PRECISION with 'auto' resolved against the autocast state."""
    if PRECISION == 'auto':
        if not torch.is_autocast_enabled('cuda'):
            return 'fp32'
        return 'fp16' if torch.get_autocast_dtype('cuda') == torch.float16 else 'bf16'
    if PRECISION not in ('fp32', 'bf16', 'fp16', 'bf16x3'):
        raise ValueError('TTRAP_PRECISION / ops.PRECISION must be auto, fp32, bf16x3, bf16 or fp16, got %r' % (PRECISION,))
    return PRECISION


def wide_storage():
    p = precision()
    mode = WIDE_STORAGE or (p if p in ('bf16', 'fp16') else 'fp32')
    if mode not in ('fp32', 'bf16', 'fp16'):
        raise ValueError('TTRAP_WIDE_STORAGE / ops.WIDE_STORAGE must be fp32, bf16 or fp16, got %r' % (mode,))
    return mode


def cl16_mode():
    """True where the layers run on 16-bit channels-last activations (either element type)."""
    return wide_storage() in ('bf16', 'fp16')


def cl16_dtype():
    """Element type of the 16-bit channels-last activations a forward creates from fp32 inputs in the current mode."""
    return torch.float16 if wide_storage() == 'fp16' else torch.bfloat16


class _HalfLib:
    """The fp16 twins of the 16-bit entry points (include/ttrap.h "fp16 twins"): attribute tt_x resolves to tt_x_h."""

    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        return getattr(self._lib, name + '_h' if name in _hip.HALF_TWINS else name)


def lib16(t):
    """The library as seen by a 16-bit channels-last tensor (or dtype) ``t``: bf16 -> the plain entry points, fp16 -> the _h twins."""
    dtype = t if isinstance(t, torch.dtype) else t.dtype
    if dtype == torch.float16:
        return _HalfLib(_hip.lib())
    if dtype != torch.bfloat16:
        raise TypeError('16-bit channels-last kernels take bfloat16 or float16 tensors, got %s' % (dtype,))
    return _hip.lib()


# Static loss scale of the fp16 channels-last backward (power of two; 1 = off).  The reference runs its train step under
# ``torch.autocast('cuda')`` = float16 WITHOUT a GradScaler (experiments/train.py:415): the loss is a mean over B x T frames, so the
# activation gradients of the first encoder levels are ~1e-7 at training batch sizes -- below fp16's normal range (6.1e-5), where every
# halving costs a bit (round 4 measured parameter gradients 3.4e-2 median / 0.53 worst off the fp32 path at 64 clips).  Here every
# activation gradient that ENTERS the fp16 region (the backward of Decoder.convout, of Encoder.convlat, an fp32 gradient arriving at a
# 16-bit layer) is multiplied by S, and everything that LEAVES it (weight / bias gradients, the gradient of the latents, of the
# encoder's input coefficients, of the skip weights) by 1 / S inside the kernels' own fp32 epilogues (include/ttrap.h:
# tt_set_loss_scale) -- an exact identity in real arithmetic, invisible to the caller: ``.grad`` holds the true gradient, no scaler
# object, the unmodified train.py benefits.  An overflow (inf / NaN gradient norm) makes FusedAdamW skip the step, as GradScaler would.
# bf16 has fp32's exponent range and is never scaled.
FP16_LOSS_SCALE = float(os.environ.get('TTRAP_FP16_LOSS_SCALE', '4096'))


def loss_scale(dtype):
    """The factor carried by the 16-bit activation gradients of element type ``dtype`` during backward."""
    if dtype != torch.float16 or FP16_LOSS_SCALE == 1.0:
        return 1.0
    s = float(FP16_LOSS_SCALE)
    if not (1.0 <= s <= 2.0 ** 24) or math.frexp(s)[0] != 0.5:
        raise ValueError('TTRAP_FP16_LOSS_SCALE / ops.FP16_LOSS_SCALE must be a power of two in [1, 2^24], got %r' % (FP16_LOSS_SCALE,))
    return s


class loss_scaled:
    """``with loss_scaled(dtype):`` around the BACKWARD calls of the 16-bit layers: sets the calling thread's loss scale in the library
    (thread-local there: autograd runs backward on its own threads) and restores the previous value."""

    def __init__(self, dtype):
        self.s = loss_scale(dtype)
        self.prev = None

    def __enter__(self):
        if self.s != 1.0:
            self.prev = _hip.lib().tt_set_loss_scale(self.s)
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            _hip.lib().tt_set_loss_scale(self.prev)
        return False


def _flags():
    # the fp32-tensor kernels (shapes without a 16-bit channels-last kernel) know bf16 operand rounding only: fp16 mode takes it too
    return {'fp32': 0, 'bf16': 1, 'fp16': 1, 'bf16x3': 2}[precision()]


@dataclass(frozen=True)
class ConvCfg:
    """Geometry of one layer. kind 'conv' = nn.Conv2d, 'tconv' = nn.ConvTranspose2d (H only)."""
    KH: int
    KW: int
    stride: int = 1
    dil: int = 1
    pad_h: int = 0
    pad_w: int = 0
    kind: str = 'conv'
    out_pad: int = 0
    act: int = ACT_NONE


def _off(t, elements):
    return ctypes.c_void_p(t.data_ptr() + 4 * elements)


def _f32c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class ConvFn(torch.autograd.Function):
    """y = act(conv(x, w) + b) for every non-fused layer of the autoencoder."""

    @staticmethod
    def forward(ctx, x, w, b, cfg):
        _hip.require_cuda(x, w)
        x, w = _f32c(x), _f32c(w)
        B, Cin, Hin, T = x.shape
        KH, KW = cfg.KH, cfg.KW
        lib = _hip.lib()
        if cfg.kind == 'conv':
            Cout = w.size(0)
            Hout = (Hin + 2 * cfg.pad_h - cfg.dil * (KH - 1) - 1) // cfg.stride + 1
            y = torch.empty((B, Cout, Hout, T), dtype=torch.float32, device=x.device)
            check(lib.tt_conv2d(ptr(x), ptr(w), ptr(b), None, ptr(y), B, Cin, Hin, T, Cout, Hout, KH, KW,
                                cfg.stride, cfg.dil, cfg.dil, cfg.pad_h, cfg.pad_w, 0,
                                Cin * KH * KW, KH * KW, KW, 1, cfg.act, stream_ptr()), 'tt_conv2d')
        else:
            Cout = w.size(1)
            Hout = (Hin - 1) * cfg.stride + KH + cfg.out_pad
            y = torch.empty((B, Cout, Hout, T), dtype=torch.float32, device=x.device)
            check(lib.tt_conv2d(ptr(x), ptr(w), ptr(b), None, ptr(y), B, Cin, Hin, T, Cout, Hout, KH, KW,
                                cfg.stride, 1, 1, 0, 0, 1,
                                KH * KW, Cout * KH * KW, KW, 1, cfg.act, stream_ptr()), 'tt_conv2d(T)')
        ctx.cfg = cfg
        ctx.has_bias = b is not None
        ctx.params = (w, b)
        ctx.save_for_backward(x, w, y if cfg.act == ACT_ELU else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        cfg = ctx.cfg
        lib = _hip.lib()
        st = stream_ptr()
        dy = _f32c(dy)
        B, Cin, Hin, T = x.shape
        _, Cout, Hout, _ = dy.shape
        KH, KW = cfg.KH, cfg.KW
        if cfg.act == ACT_ELU:
            g = torch.empty_like(dy)
            check(lib.tt_elu_bwd(ptr(dy), ptr(y), ptr(g), dy.numel(), st), 'tt_elu_bwd')
        else:
            g = dy
        dx = dw = db = rw = rb = None
        want_w, want_b = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if cfg.kind == 'conv' and cfg.stride == 1:
                # data gradient of a unit-stride conv = the same conv with both kernel axes flipped
                check(lib.tt_conv2d(ptr(g), _off(w, (KH - 1) * KW + (KW - 1)), None, None, ptr(dx),
                                    B, Cout, Hout, T, Cin, Hin, KH, KW, 1, cfg.dil, cfg.dil,
                                    (KH - 1) * cfg.dil - cfg.pad_h, (KW - 1) * cfg.dil - cfg.pad_w, 0,
                                    KH * KW, Cin * KH * KW, -KW, -1, ACT_NONE, st), 'tt_conv2d(dgrad)')
            elif cfg.kind == 'conv':
                # strided conv: gradient is the transposed form (only KW == 1 / dil == 1 layers are strided)
                check(lib.tt_conv2d(ptr(g), _off(w, KW - 1), None, None, ptr(dx),
                                    B, Cout, Hout, T, Cin, Hin, KH, KW, cfg.stride, 1, 1,
                                    cfg.pad_h, (KW - 1) - cfg.pad_w, 1,
                                    KH * KW, Cin * KH * KW, KW, -1, ACT_NONE, st), 'tt_conv2d(dgradT)')
            else:
                # transposed conv: gradient is the plain strided conv with the roles of the channel dims swapped
                check(lib.tt_conv2d(ptr(g), ptr(w), None, None, ptr(dx),
                                    B, Cout, Hout, T, Cin, Hin, KH, KW, cfg.stride, 1, 1, 0, 0, 0,
                                    Cout * KH * KW, KH * KW, KW, 1, ACT_NONE, st), 'tt_conv2d(dgrad of T)')
        # weight and bias gradients are requested independently (a frozen weight with a trainable bias still gets db)
        if want_w:
            dw, rw = _grad_target(ctx.params[0])
        if want_b:
            db, rb = _grad_target(ctx.params[1])
        if want_w:
            if cfg.kind == 'conv':
                check(lib.tt_conv2d_wgrad(ptr(x), ptr(g), ptr(dw), ptr(db), B, Cin, Hin, T, Cout, Hout, KH, KW,
                                          cfg.stride, cfg.dil, cfg.dil, cfg.pad_h, cfg.pad_w,
                                          Cin * KH * KW, KH * KW, KW, 1, st), 'tt_conv2d_wgrad')
                db = None                                   # produced by the same launch
            else:
                # dW[ci][co][kh] = sum x[ci][hi] * g[co][stride*hi + kh]: same kernel, roles swapped
                check(lib.tt_conv2d_wgrad(ptr(g), ptr(x), ptr(dw), None, B, Cout, Hout, T, Cin, Hin, KH, KW,
                                          cfg.stride, 1, 1, 0, 0,
                                          Cout * KH * KW, KH * KW, KW, 1, st), 'tt_conv2d_wgrad(T)')
        if db is not None:
            check(lib.tt_channel_sum(ptr(g), ptr(db), B, Cout, Hout * T, st), 'tt_channel_sum')
        return dx, rw, rb, None


def conv(x, w, b, cfg, link=None):
    same3 = (cfg.kind == 'conv' and (cfg.KH, cfg.KW, cfg.stride, cfg.dil, cfg.pad_h, cfg.pad_w) == (3, 3, 1, 1, 1, 1) and b is not None
             and x.dim() == 4 and x.size(3) % 2 == 0 and FUSED_RESBLOCK and x.size(2) * x.size(3) * 4 < 2 ** 31)
    if same3 and cfg.act == ACT_ELU and w.shape == (4, 2, 3, 3) and not is_cl16(x) and cl16_mode():
        return ConvIn16Fn.apply(x, w, b, link)                   # Encoder.convin feeding the bf16 channels-last interior
    if same3 and cfg.act == ACT_NONE and w.shape == (2, 4, 3, 3) and is_cl16(x):
        return ConvOut16Fn.apply(x, w, b)                        # Decoder.convout leaving it
    x = to_planar32(x)
    return ConvFn.apply(x, w, b, cfg)


class AddFn(torch.autograd.Function):
    """y = a + b (residual / skip joins)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _f32c(a), _f32c(b)
        y = torch.empty_like(a)
        check(_hip.lib().tt_scaled_add(ptr(a), ptr(b), None, 0, ptr(y), a.numel(), stream_ptr()), 'tt_scaled_add')
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class ScaleFn(torch.autograd.Function):
    """y = s[idx] * e  (TimbreTrap.apply_skip_connections, reference modules.py:112)."""

    @staticmethod
    def forward(ctx, e, s, idx):
        e, s = _f32c(e), _f32c(s)
        y = torch.empty_like(e)
        check(_hip.lib().tt_scaled_add(None, ptr(e), ptr(s), idx, ptr(y), e.numel(), stream_ptr()), 'tt_scaled_add')
        ctx.idx = idx
        ctx.save_for_backward(e, s)
        return y

    @staticmethod
    def backward(ctx, dy):
        e, s = ctx.saved_tensors
        dy = _f32c(dy)
        lib, st = _hip.lib(), stream_ptr()
        de = ds = None
        if ctx.needs_input_grad[0]:
            de = torch.empty_like(e)
            check(lib.tt_scaled_add(None, ptr(dy), ptr(s), ctx.idx, ptr(de), e.numel(), st), 'tt_scaled_add')
        if ctx.needs_input_grad[1]:
            ds = torch.zeros_like(s)
            check(lib.tt_dot(ptr(dy), ptr(e), _off(ds, ctx.idx), e.numel(), st), 'tt_dot')
        return de, ds, None


def _grad_target(p):
    """
    Where a backward kernel accumulates the gradient of parameter ``p``: (buffer, value returned to autograd).
    FusedAdamW keeps every ``.grad`` as a view of ONE flat fp32 buffer and tags its parameters; their kernels then add
    straight into that view (all weight-gradient entry points accumulate, +=) and autograd gets ``None`` -- no zero-fill
    and no ``grad += new`` launch per parameter use (~740 five-microsecond launches per train step).  Any other tensor
    gets a fresh zero buffer that is returned as usual.
    """
    if not p.requires_grad:
        # a parameter frozen after the optimizer tagged it: the kernels still need somewhere to write, but nothing may reach the
        # flat gradient buffer (it would be clipped, averaged and applied) and autograd wants no gradient for it
        return torch.zeros(p.shape, dtype=torch.float32, device=p.device), None
    g = p.grad if getattr(p, '_ttrap_accumulate', False) else None
    if g is not None and g.dtype == torch.float32 and g.is_contiguous() and g.shape == p.shape and g.device == p.device:
        p._ttrap_touched = True           # FusedAdamW: this slot has a gradient of the current step (utils/optim.py, _nograd)
        return g, None
    z = torch.zeros(p.shape, dtype=torch.float32, device=p.device)
    return z, z


class ResBlockFn(torch.autograd.Function):
    """Fused ResidualConv2dBlock (reference modules.py:755-777); hidden activations recomputed in backward."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, dilation):
        _hip.require_cuda(x, w1)
        x = _f32c(x)
        B, C, H, T = x.shape
        y = torch.empty_like(x)
        needs_grad = any(ctx.needs_input_grad[:5])
        h1 = torch.empty_like(x) if (needs_grad and SAVE_HIDDEN) else None
        with _hip.timed('resblock_fwd_C%d' % C):          # the kernel launch alone (bench.py: roofline of the dominant kernel)
            check(_hip.lib().tt_resblock_fwd(ptr(x), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(y), ptr(h1), B, C, H, T, dilation,
                                             _flags(), stream_ptr()), 'tt_resblock_fwd')
        ctx.dilation = dilation
        ctx.flags = _flags()
        ctx.params = (w1, b1, w2, b2)
        ctx.save_for_backward(x, w1, b1, w2, b2, h1)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w1, b1, w2, b2, h1 = ctx.saved_tensors
        dy = _f32c(dy)
        B, C, H, T = x.shape
        dx = torch.empty_like(x)
        (dw1, r1), (db1, r2), (dw2, r3), (db2, r4) = (_grad_target(t) for t in ctx.params)
        ws = torch.empty(x.numel() + _hip.lib().tt_wgrad_scratch_floats(), dtype=torch.float32, device=x.device)
        with _hip.timed('resblock_bwd_C%d' % C):
            check(_hip.lib().tt_resblock_bwd(ptr(x), ptr(h1), ptr(dy), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(dx), ptr(dw1),
                                             ptr(db1), ptr(dw2), ptr(db2), ptr(ws), B, C, H, T, ctx.dilation,
                                             ctx.flags, stream_ptr()), 'tt_resblock_bwd')
        return dx, r1, r2, r3, r4, None


class StridedConvFn(torch.autograd.Function):
    """EncoderBlock.sconv: ELU(Conv2d(C, 2C, (4,1), stride (2,1))) on the MFMA strided kernel."""

    @staticmethod
    def forward(ctx, x, w, b):
        _hip.require_cuda(x, w)
        x, w = _f32c(x), _f32c(w)
        B, C, H, T = x.shape
        y = torch.empty((B, 2 * C, (H - 4) // 2 + 1, T), dtype=torch.float32, device=x.device)
        check(_hip.lib().tt_sconv_fwd(ptr(x), ptr(w), ptr(b), ptr(y), B, C, H, T, stream_ptr()), 'tt_sconv_fwd')
        ctx.params = (w, b)
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = _f32c(dy)
        B, C, H, T = x.shape
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        (dw, rw), (db, rb) = (_grad_target(t) for t in ctx.params)
        scratch = torch.empty(_hip.lib().tt_wgrad_scratch_floats() + dy.numel(), dtype=torch.float32, device=x.device)
        check(_hip.lib().tt_sconv_bwd(ptr(x), ptr(y), ptr(dy), ptr(w), ptr(dx), ptr(dw), ptr(db), ptr(scratch), B, C, H, T,
                                      stream_ptr()), 'tt_sconv_bwd')
        return dx, rw, rb


class TransposedConvFn(torch.autograd.Function):
    """DecoderBlock.tconv: ELU(ConvTranspose2d(2C, C, (4,1), stride (2,1), output_padding)) on the MFMA kernel."""

    @staticmethod
    def forward(ctx, x, w, b, out_pad):
        _hip.require_cuda(x, w)
        x, w = _f32c(x), _f32c(w)
        B, C2, H, T = x.shape
        C = C2 // 2
        y = torch.empty((B, C, 2 * H + 2 + out_pad, T), dtype=torch.float32, device=x.device)
        check(_hip.lib().tt_tconv_fwd(ptr(x), ptr(w), ptr(b), ptr(y), B, C, H, T, out_pad, stream_ptr()), 'tt_tconv_fwd')
        ctx.out_pad = out_pad
        ctx.params = (w, b)
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = _f32c(dy)
        B, C2, H, T = x.shape
        C = C2 // 2
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        (dw, rw), (db, rb) = (_grad_target(t) for t in ctx.params)
        scratch = torch.empty(_hip.lib().tt_wgrad_scratch_floats() + dy.numel(), dtype=torch.float32, device=x.device)
        check(_hip.lib().tt_tconv_bwd(ptr(x), ptr(y), ptr(dy), ptr(w), ptr(dx), ptr(dw), ptr(db), ptr(scratch), B, C, H, T,
                                      ctx.out_pad, stream_ptr()), 'tt_tconv_bwd')
        return dx, rw, rb, None


def _stride16_ok(C, T, w, b):
    return (FUSED_RESBLOCK and C in WIDE_CHANNELS and (C != 4 or T % 2 == 0) and w.shape == (2 * C, C, 4, 1) and b is not None)


def strided_conv(x, w, b, win, hop, out_x3=False, link=None):
    """Conv2d(C, Cout, (win,1), stride (hop,1)) + ELU.  x may be an x3 tensor (is_x3); out_x3: the next layer takes one."""
    if is_x3(x):
        C = x.size(4)
        if (win == 4 and hop == 2 and x.size(1) >= 4 and C in X3_CHANNELS and w.shape == (2 * C, C, 4, 1) and b is not None
                and _x3_size_ok(x.size(0), x.size(1), x.size(2))):
            return x3_strided_conv(x, w, b, out_x3)
        x = from_x3(x)
    C = x.size(1)
    if (win == 4 and hop == 2 and x.size(2) >= 4 and _stride16_ok(C, x.size(-1), w, b) and (is_cl16(x) or cl16_mode())
            and _i32_ok(x, 2 * C)):
        return SConv16Fn.apply(to_cl16(x), w, b, link)
    x = to_planar32(x)
    if (out_x3 and x3_chain() and C == 8 and 2 * C in X3_CHANNELS and win == 4 and hop == 2 and x.size(2) >= 4 and x.is_cuda
            and w.shape == (2 * C, C, 4, 1) and b is not None and _x3_size_ok(x.size(0), x.size(2), x.size(3))):
        return x3_strided_conv(x, w, b, True)                    # the layer that enters the split-operand part of the encoder
    if FUSED_RESBLOCK and win == 4 and hop == 2 and C in FUSED_CHANNELS and w.shape == (2 * C, C, 4, 1) and b is not None:
        return StridedConvFn.apply(x, w, b)
    return conv(x, w, b, ConvCfg(win, 1, hop, 1, 0, 0, 'conv', 0, ACT_ELU))


def transposed_conv(x, w, b, win, hop, out_pad, out_x3=False, link=None, uplink=None):
    """ConvTranspose2d(Cin, C, (win,1), stride (hop,1), output_padding (out_pad,0)) + ELU.  x may be an x3 tensor."""
    C = w.size(1)
    if is_x3(x):
        if (win == 4 and hop == 2 and C in (16, 32) and x.size(4) == 2 * C and w.shape == (2 * C, C, 4, 1) and b is not None
                and out_pad in (0, 1) and _x3_size_ok(x.size(0), 2 * x.size(1) + 3, x.size(2))):
            return x3_transposed_conv(x, w, b, out_pad, out_x3)
        x = from_x3(x)
    if (win == 4 and hop == 2 and x.size(1) == 2 * C and out_pad in (0, 1) and _stride16_ok(C, x.size(-1), w, b)
            and (is_cl16(x) or cl16_mode()) and (2 * x.size(2) + 2 + out_pad) * x.size(3) * 2 * C < 2 ** 31):
        return TConv16Fn.apply(to_cl16(x), w, b, out_pad, link, uplink if is_cl16(x) else None)
    x = to_planar32(x)
    if (out_x3 and x3_chain() and C == 32 and x.size(1) == 64 and win == 4 and hop == 2 and x.is_cuda and w.shape == (64, 32, 4, 1)
            and b is not None and out_pad in (0, 1) and _x3_size_ok(x.size(0), 2 * x.size(2) + 3, x.size(3))):
        return x3_transposed_conv(x, w, b, out_pad, True)        # the layer that enters the split-operand part of the decoder
    if (FUSED_RESBLOCK and win == 4 and hop == 2 and C in FUSED_CHANNELS and w.shape == (2 * C, C, 4, 1)
            and x.size(1) == 2 * C and b is not None and out_pad in (0, 1)):
        return TransposedConvFn.apply(x, w, b, out_pad)
    return conv(x, w, b, ConvCfg(win, 1, hop, 1, 0, 0, 'tconv', out_pad, ACT_ELU))


def residual_block(x, w1, b1, w2, b2, dilation):
    x = to_planar32(x)
    C = x.size(1)
    if (FUSED_RESBLOCK and C in FUSED_CHANNELS and w1.shape == (C, C, 3, 3) and w2.shape == (C, C, 1, 1)
            and 1 <= dilation <= 3):
        return ResBlockFn.apply(x, w1, b1, w2, b2, dilation)
    k = w1.size(-1)
    h = conv(x, w1, b1, ConvCfg(k, k, 1, dilation, dilation * (k - 1) // 2, dilation * (k - 1) // 2, 'conv', 0, ACT_ELU))
    h = conv(h, w2, b2, ConvCfg(1, 1, 1, 1, 0, 0, 'conv', 0, ACT_ELU))
    return AddFn.apply(h, x)


# ---- bf16 channels-last activations ("cl16") ---------------------------------------------------------------------------
# In the bf16 mode the activations between the layers of the autoencoder are torch.bfloat16 tensors of LOGICAL shape
# (B,C,H,T) in torch's channels_last memory format, i.e. stored [B][H][T][C] -- what csrc/conv_wide_bf16.hip and
# csrc/conv_stride_bf16.hip read and write.  (Under autocast the reference's activations are half-precision tensors of the same
# logical shape.)  Layers that have no bf16 kernel (the 3x3 boundary convolutions, the latent heads, the losses) see fp32
# planar tensors through to_planar32 / to_cl16, which are differentiable layout changes.

CL16_CHANNELS = (4, 8, 16, 32, 64)


def is_cl16(x):
    return (x.dtype in (torch.bfloat16, torch.float16) and x.dim() == 4 and x.stride(1) == 1 and x.stride(3) == x.size(1)
            and x.stride(2) == x.size(1) * x.size(3) and x.stride(0) == x.size(1) * x.size(2) * x.size(3))


def new_cl16(B, C, H, T, device, dtype=torch.bfloat16):
    return torch.empty((B, H, T, C), dtype=dtype, device=device).permute(0, 3, 1, 2)


def _pack(x32, dtype=torch.bfloat16):
    B, C, H, T = x32.shape
    out = new_cl16(B, C, H, T, x32.device, dtype)
    check(lib16(dtype).tt_wide_pack(ptr(x32), ptr(out), B, C, H, T, stream_ptr()), 'tt_wide_pack')
    return out


def _unpack(x16):
    B, C, H, T = x16.shape
    out = torch.empty((B, C, H, T), dtype=torch.float32, device=x16.device)
    check(lib16(x16).tt_wide_unpack(ptr(x16), ptr(out), B, C, H, T, stream_ptr()), 'tt_wide_unpack')
    return out


def _cl16_ok(C, T):
    return C in CL16_CHANNELS and (C != 4 or T % 2 == 0)


def _i32_ok(x, channels=None):
    """The bf16 kernels index one clip with 32-bit offsets (shape_ok / ok_shape / edge_ok in csrc/): H * T * channels < 2^31.
    Longer one-shot inputs take the fp32 planar path like every other unsupported shape instead of raising TT_E_BADARG."""
    return x.size(2) * x.size(3) * (channels or x.size(1)) < 2 ** 31


def _as_cl16(t, dtype=torch.bfloat16):
    """Any (B,C,H,T) tensor as cl16 of element type ``dtype`` (no autograd): used on incoming gradients."""
    if is_cl16(t) and t.dtype == dtype:
        return t
    if t.dtype == torch.float32:
        # an fp32 gradient ENTERS the 16-bit region here: it takes the loss scale of the region (1 for bf16) before it is rounded
        s = loss_scale(dtype)
        t = t if s == 1.0 else t * s
        if _cl16_ok(t.size(1), t.size(3)):
            return _pack(t.contiguous(), dtype)
    return t.to(dtype).contiguous(memory_format=torch.channels_last)


class ToCL16Fn(torch.autograd.Function):
    """fp32 planar (B,C,H,T) -> cl16; the gradient comes back as fp32 planar."""

    @staticmethod
    def forward(ctx, x):
        _hip.require_cuda(x)
        ctx.dtype = cl16_dtype()
        return _pack(_f32c(x), ctx.dtype)

    @staticmethod
    def backward(ctx, g):
        out = _unpack(_as_cl16(g, ctx.dtype))
        s = loss_scale(ctx.dtype)                    # the gradient LEAVES the 16-bit region: the loss scale comes off
        return out if s == 1.0 else out.mul_(1.0 / s)


class ToPlanar32Fn(torch.autograd.Function):
    """cl16 -> fp32 planar (B,C,H,T); the gradient goes back as cl16."""

    @staticmethod
    def forward(ctx, x):
        ctx.dtype = x.dtype
        return _unpack(x)

    @staticmethod
    def backward(ctx, g):
        s = loss_scale(ctx.dtype)                    # the gradient ENTERS the 16-bit region
        return _pack(_f32c(g) if s == 1.0 else _f32c(g) * s, ctx.dtype)


def to_cl16(x):
    if is_cl16(x):
        return x
    if not _cl16_ok(x.size(1), x.size(3)):
        raise ValueError('no bf16 channels-last form for %d channels x %d frames' % (x.size(1), x.size(3)))
    return ToCL16Fn.apply(x)


def to_planar32(x):
    """What every fp32-only layer calls on its input: identity for fp32 tensors."""
    if is_cl16(x):
        return ToPlanar32Fn.apply(x)
    if is_x3(x):
        return from_x3(x)
    return x


class ConvIn16Fn(torch.autograd.Function):
    """Encoder.convin (3x3, 2 -> 4, ELU): fp32 planar coefficients -> cl16 (csrc/conv_edge_bf16.hip)."""

    @staticmethod
    def forward(ctx, x, w, b, link=None):
        _hip.require_cuda(x, w)
        x = _f32c(x)
        B, _, H, T = x.shape
        y = new_cl16(B, 4, H, T, x.device, cl16_dtype())
        check(lib16(y).tt_convin16_fwd(ptr(x), ptr(w), ptr(b), ptr(y), B, H, T, stream_ptr()), 'tt_convin16_fwd')
        ctx.params = (w, b)
        ctx.link = link                                          # GateLink with the first level (its backward may hand dy back gated)
        if link is not None:
            link.producer = True
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        B, _, H, T = x.shape
        lib = lib16(y)
        g = _as_cl16(dy, y.dtype)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        (dw, r1), (db, r2) = (_grad_target(t) for t in ctx.params)
        ws = torch.empty(lib.tt_edge16_scratch_bytes(), dtype=torch.uint8, device=x.device)
        pre = ctx.link is not None and ctx.link.gated           # dy arrives as dy * ELU'(y): y is not read
        with loss_scaled(y.dtype):
            check(lib.tt_convin16_bwd(ptr(x), None if pre else ptr(y), ptr(g), ptr(w), ptr(dx), ptr(dw), ptr(db), ptr(ws), B, H, T, stream_ptr()),
                  'tt_convin16_bwd')
        return dx, r1, r2, None


class ConvOut16Fn(torch.autograd.Function):
    """Decoder.convout (3x3, 4 -> 2, no activation): cl16 -> fp32 planar logits."""

    @staticmethod
    def forward(ctx, x, w, b):
        B, _, H, T = x.shape
        y = torch.empty((B, 2, H, T), dtype=torch.float32, device=x.device)
        check(lib16(x).tt_convout16_fwd(ptr(x), ptr(w), ptr(b), ptr(y), B, H, T, stream_ptr()), 'tt_convout16_fwd')
        ctx.params = (w, b)
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        B, _, H, T = x.shape
        lib = lib16(x)
        dy = _f32c(dy)
        dx = new_cl16(B, 4, H, T, x.device, x.dtype)
        (dw, r1), (db, r2) = (_grad_target(t) for t in ctx.params)
        ws = torch.empty(lib.tt_edge16_scratch_bytes(), dtype=torch.uint8, device=x.device)
        with loss_scaled(x.dtype):                      # dx enters the 16-bit region scaled; dw, db come from the fp32 dy itself
            check(lib.tt_convout16_bwd(ptr(x), ptr(dy), ptr(w), ptr(dx), ptr(dw), ptr(db), ptr(ws), B, H, T, stream_ptr()),
                  'tt_convout16_bwd')
        return dx, r1, r2


class ConvOut16PairFn(torch.autograd.Function):
    """Decoder.convout on a batch that is two batches back to back (TimbreTrap.decode_pair: the reconstruction and the transcription
    decode of the same latents in ONE pass through the decoder): returns the two halves as two tensors of their own, so that the losses'
    gradients come back as two tensors as well -- a sliced single output would cost autograd a zero-filled full-size gradient and an
    add per slice (0.55 ms per step measured for such slices, bench.py)."""

    @staticmethod
    def forward(ctx, x, w, b):
        B, _, H, T = x.shape
        h = B // 2
        lib, st = lib16(x), stream_ptr()
        ys = [torch.empty((h, 2, H, T), dtype=torch.float32, device=x.device) for _ in range(2)]
        for i, y in enumerate(ys):
            check(lib.tt_convout16_fwd(ptr(x[i * h:(i + 1) * h]), ptr(w), ptr(b), ptr(y), h, H, T, st), 'tt_convout16_fwd')
        ctx.params = (w, b)
        ctx.save_for_backward(x, w)
        return ys[0], ys[1]

    @staticmethod
    def backward(ctx, g0, g1):
        x, w = ctx.saved_tensors
        B, _, H, T = x.shape
        h = B // 2
        lib, st = lib16(x), stream_ptr()
        dx = new_cl16(B, 4, H, T, x.device, x.dtype)
        (dw, r1), (db, r2) = (_grad_target(t) for t in ctx.params)
        ws = torch.empty(lib.tt_edge16_scratch_bytes(), dtype=torch.uint8, device=x.device)
        with loss_scaled(x.dtype):
            for i, g in enumerate((g0, g1)):
                if g is None:
                    dx[i * h:(i + 1) * h].zero_()
                    continue
                check(lib.tt_convout16_bwd(ptr(x[i * h:(i + 1) * h]), ptr(_f32c(g)), ptr(w), ptr(dx[i * h:(i + 1) * h]), ptr(dw), ptr(db), ptr(ws),
                                           h, H, T, st), 'tt_convout16_bwd')
        return dx, r1, r2


def conv_out_pair(x, w, b):
    """Conv2d(4, 2, 3, padding 'same') on a batch of two halves -> (first half, second half) (ConvOut16PairFn; else conv + slices)."""
    if (x.dim() == 4 and x.size(0) % 2 == 0 and is_cl16(x) and w.shape == (2, 4, 3, 3) and b is not None and x.size(3) % 2 == 0
            and FUSED_RESBLOCK and x.size(2) * x.size(3) * 4 < 2 ** 31):
        return ConvOut16PairFn.apply(x, w, b)
    y = conv(x, w, b, ConvCfg(3, 3, 1, 1, 1, 1, 'conv', 0, ACT_NONE))
    h = y.size(0) // 2
    return y[:h], y[h:]


# (Clips-per-pass chunking of a level -- three blocks back to back on a few clips so that a block finds its input in the 256 MB memory-side
# cache -- was measured in rounds 2 and 4 and removed in round 6: 69.9 ms per step unchunked, 74.1 / 79.8 / 93.6 ms with chunks of 32 / 16 / 8
# clips; smaller launches cost more than the cache gives.)


# TTRAP_LEVEL_RECOMPUTE=1: the residual blocks of the wide levels (C = 16, 32) run their backward as ONE fused pass that
# recomputes the hidden activation per tile (csrc/conv_level_bf16.hip, tt_wide_rb_bwd_fused): the forward stores no h1 and
# dL/d(conv1 pre-activation) never reaches HBM -- 5 tensor passes per block (x, y | x, dy, dx) instead of 11, and a third less saved
# activation memory at those levels (52.6 -> 42.7 GB peak for the 64-clip step).  Measured (round 3): the fused pass is bound by
# vector-ALU issue, not by HBM (0.52-1.1 ms per block against 0.47-0.52 ms for the three per-stage kernels; step 69.3 -> 76.6 ms),
# so it is the memory-saving option, not the default.
RECOMPUTE_CHANNELS = (16, 32) if os.environ.get('TTRAP_LEVEL_RECOMPUTE', '0') == '1' else ()


# TTRAP_LEVEL_BWD=0: one tt_wide_rb_bwd call (with its own reduce launch) per block instead of tt_wide_level_bwd (A/B switch)
LEVEL_BWD = os.environ.get('TTRAP_LEVEL_BWD', '1') != '0'


# A residual level's backward can hand the layer in front of it its gradient ALREADY multiplied by that layer's ELU derivative: the
# level's input IS that layer's output (modules.py:683-693: tconv + ELU -> block1), and the first block's data-gradient kernel has it in
# LDS when it writes dx (tt_wide_level_bwd_gated).  The layer's backward then skips reading its saved output and stages nothing through
# registers (tt_tconv16_bwd_pregated / tt_sconv16_bwd_pregated).  Both sides must agree, and EVERY gradient that reaches the producing
# layer must carry the factor: a GateLink is created by the module that owns both calls (DecoderBlock: the intermediate tensor has no
# other consumer; Encoder: the strided layer's output is also an embedding handed to the caller -- that copy goes through gate_tap,
# whose backward applies the factor to whatever gradient comes back through it, skip connections included).  The producer marks the
# link in its forward when it took the 16-bit path; the level, in ITS forward, promises to gate when it will run tt_wide_level_bwd
# (link.gated); the producer's backward and the tap's read the promise.  TTRAP_PREGATE=0 / ops.PREGATE = False: never (A/B).
PREGATE = os.environ.get('TTRAP_PREGATE', '1') != '0'


# The same between the latent heads and their neighbours: Encoder.convlat's data gradient leaves gated for the last strided layer
# (tt_latent16_expand_gated), and the first DecoderBlock's transposed layer gates ITS dx for Decoder.convin -- which it can only do in its
# pregated form, i.e. when the level behind it gates in turn: that promise `depends` on the other link's, read at backward time.
class GateLink:
    __slots__ = ('producer', '_gated', 'depends', 'accumulates', 'pending')

    def __init__(self):
        # Deferred skip joins (round 6): `accumulates` -- set by the forward of the layer that CONSUMES the linked tensor inside the encoder
        # (Level16Fn, LatEnc16Fn): its backward will fold whatever sits in `pending` into the data gradient it writes (flush_pending);
        # `pending` -- the backward of a skip join on the same tensor (which runs earlier: the decoder comes after the encoder) parks its
        # arguments there instead of writing a gradient tensor of its own that autograd would then have to add to that data gradient.
        self.accumulates = False
        self.pending = []
        self.producer = False        # set by the producing layer's forward (SConv16Fn / TConv16Fn / LatDec16Fn): it is a 16-bit one and
                                     # its backward will look at `gated`
        self._gated = False          # set by the consumer's forward (Level16Fn / LatEnc16Fn / TConv16Fn): the gradient its backward
                                     # returns will carry the producer's ELU'
        self.depends = None          # ... provided this other link's consumer gates too

    @property
    def gated(self):
        return self._gated and (self.depends is None or self.depends.gated)

    @gated.setter
    def gated(self, v):
        self._gated = bool(v)


def gate_link():
    """A GateLink for a (strided / transposed layer -> residual level) pair whose intermediate tensor nobody else consumes, or None."""
    return GateLink() if PREGATE else None


def _level16_forward(ctx, x, dilations, link, join, params):
    """Level16Fn / Level16JoinFn forward.  ``join`` = None or (e, weights, idx, link of e): the level's LAST block adds weights[idx] * e[b mod Be]
    in its epilogue (tt_wide_rb_fwd_join)."""
    B, C, H, T = x.shape
    lib, st = lib16(x), stream_ptr()
    needs_grad = any(ctx.needs_input_grad)
    recompute = C in RECOMPUTE_CHANNELS
    # the promise to gate: only where backward will take the one-call path below (all of it known now)
    ctx.gate = bool(link is not None and link.producer and ctx.needs_input_grad[0] and LEVEL_BWD and not recompute and len(dilations) <= 4)
    if ctx.gate:
        link.gated = True
    ctx.link = link
    if link is not None and ctx.needs_input_grad[0]:
        link.accumulates = True             # this level's backward folds parked skip-join backwards into its dx (flush_pending)
    nb = len(dilations)
    outs = [new_cl16(B, C, H, T, x.device, x.dtype) for _ in range(nb)]
    hids = [new_cl16(B, C, H, T, x.device, x.dtype) if (needs_grad and not recompute) else None for _ in range(nb)]
    cur = x
    for i, d in enumerate(dilations):
        w1, b1_, w2, b2 = params[4 * i: 4 * i + 4]
        with _hip.timed('wide_rb_fwd_C%d' % C, clips=B):
            if join is not None and i == nb - 1:
                je, jw, jidx = join[:3]
                check(lib.tt_wide_rb_fwd_join(ptr(cur), ptr(w1), ptr(b1_), ptr(w2), ptr(b2), ptr(outs[i]), ptr(hids[i]), ptr(je), ptr(jw), jidx,
                                              je.size(0), B, C, H, T, d, st), 'tt_wide_rb_fwd_join')
            else:
                check(lib.tt_wide_rb_fwd(ptr(cur), ptr(w1), ptr(b1_), ptr(w2), ptr(b2), ptr(outs[i]), ptr(hids[i]), B, C, H, T, d, st), 'tt_wide_rb_fwd')
        cur = outs[i]
    ctx.dilations = tuple(dilations)
    ctx.params = params
    ctx.recompute = recompute
    ctx.join = None if join is None else (join[2], join[3], join[1], join[4])        # idx, link, the weights parameter object, defer
    if needs_grad:
        saved = []
        for i in range(nb):
            saved += [x if i == 0 else outs[i - 1]] + ([] if recompute else [hids[i]])
        ctx.save_for_backward(*params, *(() if join is None else join[:2]), *saved)
    return outs[-1]


def _level16_backward(ctx, dy):
    """-> (dx, [values returned to autograd for the parameters], de, value returned for the skip weights)."""
    nb = len(ctx.dilations)
    tensors = ctx.saved_tensors
    nj = 0 if ctx.join is None else 2
    params, saved = tensors[:4 * nb], tensors[4 * nb + nj:]
    B, C, H, T = saved[0].shape
    dt = saved[0].dtype
    lib, st = lib16(dt), stream_ptr()
    g_all = _as_cl16(dy, dt)
    de = rs = None
    if ctx.join is not None:
        # the folded join's backward: de = w * (sum over the halves of dy) [* ELU'(e)], dw += <sum, e> -- dy itself goes on into the blocks
        je, jw = tensors[4 * nb: 4 * nb + 2]
        jidx, jlink, jparam, jdefer = ctx.join
        de, rs = _join_backward(g_all, je, jw, jidx, B // je.size(0), jlink, jparam, ctx.needs_input_grad[3], ctx.needs_input_grad[4], jdefer)
    recompute = ctx.recompute
    ws_bytes = lib.tt_wide_fused_scratch_bytes(C) if recompute else lib.tt_wide_scratch_bytes(B, C, H, T)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=g_all.device)
    targets = [_grad_target(t) for t in ctx.params]
    dx = new_cl16(B, C, H, T, g_all.device, dt)
    tmp = [new_cl16(B, C, H, T, g_all.device, dt) for _ in range(2)] if nb > 1 else []
    if LEVEL_BWD and not recompute and nb <= 4:
        # the whole level in one call: the partial-sum reduces of its blocks are one launch at the end (tt_wide_level_bwd)
        def arr(ts):
            return (ctypes.c_void_p * nb)(*[t.data_ptr() for t in ts])
        ws = torch.empty(lib.tt_wide_level_scratch_bytes(nb, B, C, H, T), dtype=torch.uint8, device=g_all.device)
        cols = list(zip(*[[targets[4 * i + j][0] for j in range(4)] for i in range(nb)]))      # dw1s, db1s, dw2s, db2s
        dil = (ctypes.c_int * nb)(*ctx.dilations)
        fn = lib.tt_wide_level_bwd_gated if ctx.gate else lib.tt_wide_level_bwd
        args = (nb, arr([saved[2 * i] for i in range(nb)]), arr([saved[2 * i + 1] for i in range(nb)]), ptr(g_all),
                arr([params[4 * i] for i in range(nb)]), arr([params[4 * i + 2] for i in range(nb)]),
                arr([params[4 * i + 3] for i in range(nb)]), ptr(dx), ptr(tmp[0]) if tmp else None,
                ptr(tmp[1]) if tmp else None, arr(cols[0]), arr(cols[1]), arr(cols[2]), arr(cols[3]), ptr(ws),
                B, C, H, T, dil)
        ride = _riding_join(ctx, saved[0])
        with _hip.timed('wide_rb_bwd_C%d' % C, clips=B):
            if ride is not None:
                # one parked skip-join backward on this level's input: it rides on the first block's gated epilogue (tt_wide_level_bwd_gated_join)
                pg, pe, pw, pidx, preps, pds = ride
                rc = lib.tt_wide_level_bwd_gated_join(*args, ptr(pg), preps, ptr(pw), pidx, ptr(pds), st)
                if rc == 0:
                    ctx.link.pending = []
                elif rc != 1:                                    # 1 = TT_W_JOIN_LEFT: the level is done, the join is left to flush_pending below
                    check(rc, 'tt_wide_level_bwd_gated_join')
            else:
                check(fn(*args, st), 'tt_wide_level_bwd')
        flush_pending(ctx.link, dx)
        return dx, [r for _, r in targets], de, rs
    if ctx.gate:
        raise RuntimeError('ops.LEVEL_BWD / RECOMPUTE_CHANNELS changed between the forward and the backward of a level')
    g = g_all
    for i in reversed(range(nb)):
        w1, b1_, w2, b2 = params[4 * i: 4 * i + 4]
        (dw1, _), (db1, _), (dw2, _), (db2, _) = targets[4 * i: 4 * i + 4]
        gx = dx if i == 0 else tmp[i & 1]
        with _hip.timed('wide_rb_bwd_C%d' % C, clips=B):
            if recompute:
                check(lib.tt_wide_rb_bwd_fused(ptr(saved[i]), ptr(g), ptr(w1), ptr(b1_), ptr(w2), ptr(b2), ptr(gx), ptr(dw1), ptr(db1),
                                               ptr(dw2), ptr(db2), ptr(ws), B, C, H, T, ctx.dilations[i], st), 'tt_wide_rb_bwd_fused')
            else:
                check(lib.tt_wide_rb_bwd(ptr(saved[2 * i]), ptr(saved[2 * i + 1]), ptr(g), ptr(w1), ptr(w2), ptr(b2), ptr(gx), ptr(dw1), ptr(db1),
                                         ptr(dw2), ptr(db2), ptr(ws), B, C, H, T, ctx.dilations[i], st), 'tt_wide_rb_bwd')
        g = gx
    flush_pending(ctx.link, dx)
    return dx, [r for _, r in targets], de, rs


class Level16Fn(torch.autograd.Function):
    """The residual blocks of one level on cl16 tensors (csrc/conv_wide_bf16.hip, csrc/conv_level_bf16.hip); see WideLevelFn for
    the fp32-facing form.  Saved for backward: the input of every block, plus its hidden activation at the widths whose
    backward does not recompute it (RECOMPUTE_CHANNELS)."""

    @staticmethod
    def forward(ctx, x, dilations, link, *params):
        return _level16_forward(ctx, x, dilations, link, None, params)

    @staticmethod
    def backward(ctx, dy):
        dx, rp, _, _ = _level16_backward(ctx, dy)
        return (dx, None, None, *rp)


class Level16JoinFn(torch.autograd.Function):
    """Level16Fn whose LAST block adds the weighted skip in its epilogue: level(x) + weights[idx] * e[b mod Be] (round 6: the join behind a
    DecoderBlock, reference modules.py:569-589, without a pass of its own -- tt_wide_rb_fwd_join; backward = the level's backward on the
    incoming gradient + tt_skip_join16_bwd for e and the weight).  ``elink``: the GateLink of e (see SkipJoin16Fn)."""

    @staticmethod
    def forward(ctx, x, dilations, link, e, weights, idx, elink, defer, *params):
        return _level16_forward(ctx, x, dilations, link, (e, weights, idx, elink, defer), params)

    @staticmethod
    def backward(ctx, dy):
        dx, rp, de, rs = _level16_backward(ctx, dy)
        return (dx, None, None, de, rs, None, None, None, *rp)


class SConv16Fn(torch.autograd.Function):
    """EncoderBlock.sconv on cl16 tensors: (B,C,H,T) -> (B,2C,(H-4)/2+1,T) (csrc/conv_stride_bf16.hip)."""

    @staticmethod
    def forward(ctx, x, w, b, link=None):
        B, C, H, T = x.shape
        y = new_cl16(B, 2 * C, (H - 4) // 2 + 1, T, x.device, x.dtype)
        check(lib16(x).tt_sconv16_fwd(ptr(x), ptr(w), ptr(b), ptr(y), B, C, H, T, stream_ptr()), 'tt_sconv16_fwd')
        ctx.params = (w, b)
        ctx.link = link
        if link is not None:
            link.producer = True
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        B, C, H, T = x.shape
        lib = lib16(x)
        g = _as_cl16(dy, x.dtype)
        dx = new_cl16(B, C, H, T, x.device, x.dtype) if ctx.needs_input_grad[0] else None
        (dw, r1), (db, r2) = (_grad_target(t) for t in ctx.params)
        ws = torch.empty(lib.tt_stride16_scratch_bytes(C), dtype=torch.uint8, device=x.device)
        if ctx.link is not None and ctx.link.gated:              # the level behind this layer left dy * ELU'(y)
            check(lib.tt_sconv16_bwd_pregated(ptr(x), ptr(g), ptr(w), ptr(dx), ptr(dw), ptr(db), ptr(ws), B, C, H, T, stream_ptr()),
                  'tt_sconv16_bwd_pregated')
            return dx, r1, r2, None
        check(lib.tt_sconv16_bwd(ptr(x), ptr(y), ptr(g), ptr(w), ptr(dx), ptr(dw), ptr(db), ptr(ws), B, C, H, T, stream_ptr()),
              'tt_sconv16_bwd')
        return dx, r1, r2, None


class TConv16Fn(torch.autograd.Function):
    """DecoderBlock.tconv on cl16 tensors: (B,2C,H,T) -> (B,C,2H+2+out_pad,T)."""

    @staticmethod
    def forward(ctx, x, w, b, out_pad, link=None, uplink=None):
        B, C2, H, T = x.shape
        C = C2 // 2
        # uplink: x is the ELU output of a 16-bit layer that takes its gradient gated (LatDec16Fn); this layer's pregated backward can do it
        ctx.uplink = uplink if (uplink is not None and uplink.producer and link is not None and C in (16, 32)
                                and ctx.needs_input_grad[0]) else None
        if ctx.uplink is not None:
            uplink.depends = link
            uplink.gated = True
        y = new_cl16(B, C, 2 * H + 2 + out_pad, T, x.device, x.dtype)
        check(lib16(x).tt_tconv16_fwd(ptr(x), ptr(w), ptr(b), ptr(y), B, C, H, T, out_pad, stream_ptr()), 'tt_tconv16_fwd')
        ctx.params = (w, b)
        ctx.out_pad = out_pad
        ctx.link = link
        if link is not None:
            link.producer = True
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        B, C2, H, T = x.shape
        C = C2 // 2
        lib = lib16(x)
        g = _as_cl16(dy, x.dtype)
        dx = new_cl16(B, C2, H, T, x.device, x.dtype) if ctx.needs_input_grad[0] else None
        (dw, r1), (db, r2) = (_grad_target(t) for t in ctx.params)
        ws = torch.empty(lib.tt_stride16_scratch_bytes(C), dtype=torch.uint8, device=x.device)
        if ctx.link is not None and ctx.link.gated:              # the level behind this layer left dy * ELU'(y)
            check(lib.tt_tconv16_bwd_pregated(ptr(x), ptr(g), ptr(w), ptr(dx), ptr(dw), ptr(db), ptr(ws), B, C, H, T, ctx.out_pad,
                                              1 if ctx.uplink is not None else 0, stream_ptr()), 'tt_tconv16_bwd_pregated')
            return dx, r1, r2, None, None, None
        check(lib.tt_tconv16_bwd(ptr(x), ptr(y), ptr(g), ptr(w), ptr(dx), ptr(dw), ptr(db), ptr(ws), B, C, H, T, ctx.out_pad,
                                 stream_ptr()), 'tt_tconv16_bwd')
        return dx, r1, r2, None, None, None


_SCALARS = {}


def _device_scalar(value, device):
    """A one-element fp32 device tensor holding ``value`` (cached: the loss scale and its reciprocal as kernel scale arguments)."""
    key = (float(value), str(device))
    t = _SCALARS.get(key)
    if t is None:
        t = _SCALARS[key] = torch.full((1,), float(value), dtype=torch.float32, device=device)
    return t


class GateTapFn(torch.autograd.Function):
    """Identity on the output y of a 16-bit layer, for the copy of it that LEAVES the module (an encoder embedding handed to the caller:
    skip connections through the public apply_skip_connections / decode, a caller's own use).  Two duties in its backward, one pass:
      * the module boundary of the loss-scaled fp16 backward (FP16_LOSS_SCALE): whatever comes back through this copy is a TRUE gradient
        -- torch's own ops (``emb.float()``, a custom loss on an embedding) know nothing of the scale -- and takes the factor S here, where
        it enters the 16-bit region (round-5 verdict weak #15 / advisor: such a gradient used to be taken for a scaled one and came out
        4096x too small, silently);
      * where the layer's OTHER consumer is a level that gates (GateLink), the factor ELU'(y), so that every contribution to the layer's
        incoming gradient carries it."""

    @staticmethod
    def forward(ctx, y, link):
        ctx.link = link
        ctx.save_for_backward(y)
        return y.as_strided(y.size(), y.stride(), y.storage_offset())     # (view_as renumbers the stride of a size-1 batch dimension)

    @staticmethod
    def backward(ctx, g):
        y, = ctx.saved_tensors
        gated = ctx.link is not None and ctx.link.gated
        s = loss_scale(y.dtype)
        if not gated and s == 1.0:
            return g, None
        if g.dtype == torch.float32:
            g16, s = _as_cl16(g, y.dtype), 1.0                   # (an fp32 gradient takes the scale on its way to 16 bits)
        else:
            g16 = _as_cl16(g, y.dtype)
        out = new_cl16(*y.shape, y.device, y.dtype)
        # out = s * g * (ELU'(y) if gated): the backward of the fused skip join with one batch and no weight gradient
        check(lib16(y).tt_skip_join16_bwd(ptr(g16), ptr(y), ptr(_device_scalar(s, y.device)), 0, ptr(out), None, y.numel(), 1, int(gated),
                                          stream_ptr()), 'tt_skip_join16_bwd')
        return out, None


def gate_tap(y, link):
    """The copy of a 16-bit layer's output that leaves the module (see GateTapFn); y itself when there is nothing to do."""
    if not torch.is_grad_enabled() or not y.requires_grad or not is_cl16(y) or y.numel() % 8:
        return y
    if (link is None or not link.producer) and loss_scale(y.dtype) == 1.0:
        return y
    return GateTapFn.apply(y, link)


class WideLevelFn(torch.autograd.Function):
    """
    The residual blocks of one wide level (reference modules.py:621-624 / 690-693: block1..3, dilation 1, 2, 3) with bf16
    channel-innermost activations in HBM (csrc/conv_wide_bf16.hip).  Input and output are ordinary fp32 (B,C,H,T) tensors;
    the block inputs and hidden activations saved for backward are bf16 (half the bytes of the fp32 path).
    Arguments after x: dilations (tuple), then w1, b1, w2, b2 of every block.
    """

    @staticmethod
    def forward(ctx, x, dilations, *params):
        _hip.require_cuda(x, params[0])
        x = _f32c(x)
        B, C, H, T = x.shape
        ctx.dtype = dt = cl16_dtype()
        lib, st = lib16(dt), stream_ptr()
        nb = len(dilations)
        needs_grad = any(ctx.needs_input_grad)
        cur = torch.empty((B, H, T, C), dtype=dt, device=x.device)
        check(lib.tt_wide_pack(ptr(x), ptr(cur), B, C, H, T, st), 'tt_wide_pack')
        saved = []
        for i, d in enumerate(dilations):
            w1, b1, w2, b2 = params[4 * i: 4 * i + 4]
            nxt = torch.empty_like(cur)
            h1 = torch.empty_like(cur) if needs_grad else None
            with _hip.timed('wide_rb_fwd_C%d' % C):
                check(lib.tt_wide_rb_fwd(ptr(cur), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(nxt), ptr(h1), B, C, H, T, d, st),
                      'tt_wide_rb_fwd')
            saved += [cur, h1]
            cur = nxt
        y = torch.empty_like(x)
        check(lib.tt_wide_unpack(ptr(cur), ptr(y), B, C, H, T, st), 'tt_wide_unpack')
        ctx.dilations = tuple(dilations)
        ctx.params = params
        ctx.geom = (B, C, H, T)
        if needs_grad:
            ctx.save_for_backward(*params, *saved)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, H, T = ctx.geom
        nb = len(ctx.dilations)
        tensors = ctx.saved_tensors
        params, saved = tensors[:4 * nb], tensors[4 * nb:]
        lib, st = lib16(ctx.dtype), stream_ptr()
        dy = _f32c(dy)
        ls = loss_scale(ctx.dtype)                                # fp32 outside, 16-bit inside: the loss scale goes on here and comes off at the end
        if ls != 1.0:
            dy = dy * ls
        g = torch.empty((B, H, T, C), dtype=ctx.dtype, device=dy.device)
        check(lib.tt_wide_pack(ptr(dy), ptr(g), B, C, H, T, st), 'tt_wide_pack')
        ws = torch.empty(lib.tt_wide_scratch_bytes(B, C, H, T), dtype=torch.uint8, device=dy.device)
        grads = [None] * (4 * nb)
        for i in reversed(range(nb)):
            w1, b1, w2, b2 = params[4 * i: 4 * i + 4]
            xin, h1 = saved[2 * i], saved[2 * i + 1]
            (dw1, r1), (db1, r2), (dw2, r3), (db2, r4) = (_grad_target(t) for t in ctx.params[4 * i: 4 * i + 4])
            gx = torch.empty_like(g)
            with _hip.timed('wide_rb_bwd_C%d' % C):
                check(lib.tt_wide_rb_bwd(ptr(xin), ptr(h1), ptr(g), ptr(w1), ptr(w2), ptr(b2), ptr(gx), ptr(dw1), ptr(db1),
                                         ptr(dw2), ptr(db2), ptr(ws), B, C, H, T, ctx.dilations[i], st), 'tt_wide_rb_bwd')
            grads[4 * i: 4 * i + 4] = [r1, r2, r3, r4]
            g = gx
        dx = torch.empty((B, C, H, T), dtype=torch.float32, device=dy.device)
        check(lib.tt_wide_unpack(ptr(g), ptr(dx), B, C, H, T, st), 'tt_wide_unpack')
        if ls != 1.0:
            dx.mul_(1.0 / ls)
        return (dx, None, *grads)


class Add16Fn(torch.autograd.Function):
    """a + b on cl16 tensors (the skip joins of the 16-bit path when the skips arrive as scaled TENSORS -- the public
    apply_skip_connections / decode route; csrc/conv_generic.hip: tt_scaled_add16).  a is the decoder's own activation, b the skip: a
    tensor that crossed the module boundary, so its gradient leaves the loss-scaled region here (FP16_LOSS_SCALE: b's gradient is the
    TRUE one, divided by S; torch's own ops and Scale16Fn between here and the encoder's tap -- where S goes back on -- see true gradients)."""

    @staticmethod
    def forward(ctx, a, b):
        B, C, H, T = a.shape
        if b.dtype != a.dtype:
            raise TypeError('skip join of %s and %s tensors' % (a.dtype, b.dtype))
        y = new_cl16(B, C, H, T, a.device, a.dtype)
        check(lib16(a).tt_scaled_add16(ptr(a), ptr(b), None, 0, ptr(y), a.numel(), stream_ptr()), 'tt_scaled_add16')
        ctx.dtype = a.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        g = _as_cl16(dy, ctx.dtype)
        s = loss_scale(ctx.dtype)
        if s == 1.0 or not ctx.needs_input_grad[1]:
            return g, g
        gb = new_cl16(*g.shape, g.device, g.dtype)
        check(lib16(g).tt_scaled_add16(None, ptr(g), ptr(_device_scalar(1.0 / s, g.device)), 0, ptr(gb), g.numel(), stream_ptr()), 'tt_scaled_add16')
        return g, gb


class Scale16Fn(torch.autograd.Function):
    """s[idx] * e on a cl16 tensor (TimbreTrap.apply_skip_connections, reference modules.py:112) -- fp32 weight, 16-bit tensor.  Its
    input and output are tensors OUTSIDE the modules: true gradients in and out, no loss scale (see Add16Fn / GateTapFn)."""

    @staticmethod
    def forward(ctx, e, s, idx):
        B, C, H, T = e.shape
        s = _f32c(s)
        y = new_cl16(B, C, H, T, e.device, e.dtype)
        check(lib16(e).tt_scaled_add16(None, ptr(e), ptr(s), idx, ptr(y), e.numel(), stream_ptr()), 'tt_scaled_add16')
        ctx.idx = idx
        ctx.save_for_backward(e, s)
        return y

    @staticmethod
    def backward(ctx, dy):
        e, s = ctx.saved_tensors
        g = _as_cl16(dy, e.dtype)
        lib, st = lib16(e), stream_ptr()
        de = ds = None
        if ctx.needs_input_grad[0]:
            B, C, H, T = e.shape
            de = new_cl16(B, C, H, T, e.device, e.dtype)
            check(lib.tt_scaled_add16(None, ptr(g), ptr(s), ctx.idx, ptr(de), e.numel(), st), 'tt_scaled_add16')
        if ctx.needs_input_grad[1]:
            ds = torch.zeros_like(s)
            check(lib.tt_dot16(ptr(g), ptr(e), _off(ds, ctx.idx), e.numel(), st), 'tt_dot16')
        return de, ds, None


class SkipJoin:
    """A skip connection that has not been applied yet: (encoder embedding, the skip weights, which one, the embedding's GateLink).
    TimbreTrap.forward hands these to the decoder on the 16-bit path instead of the scaled tensors of apply_skip_connections: the
    product and the join are then ONE pass (SkipJoin16Fn) and the embedding serves both halves of a pair decode."""
    __slots__ = ('e', 'weights', 'idx', 'link', 'defer')

    def __init__(self, e, weights, idx, link=None, defer=False):
        self.e, self.weights, self.idx, self.link = e, weights, idx, link
        # defer (TimbreTrap.forward only, where encoder and decoder are one graph): the embedding's share of the join's backward may be left
        # to the backward of the encoder layer behind the embedding (GateLink.pending) -- see _join_backward
        self.defer = bool(defer)


# TTRAP_SKIP_FUSED=0 / ops.SKIP_FUSED = False: scale and join as two passes per decode, decodes one by one (the round 2-5 route; A/B)
SKIP_FUSED = os.environ.get('TTRAP_SKIP_FUSED', '1') != '0'


# TTRAP_SKIP_DEFER=0 / ops.SKIP_DEFER = False: every skip join writes the embedding's gradient in its own backward and autograd adds it to the
# encoder level's data gradient (A/B)
SKIP_DEFER = os.environ.get('TTRAP_SKIP_DEFER', '1') != '0'


def _join_backward(g, e, weights, idx, reps, link, param, want_e, want_w, defer):
    """The backward of a weighted skip join, out[r] = y[r] + weights[idx] * e (r < reps), for e and the weight; g = the incoming 16-bit
    gradient (it IS dy).  -> (de, value returned to autograd for the weights).
    Deferred form: where the embedding's other consumer is an encoder layer that accumulates (GateLink.accumulates) and the weight's
    gradient goes straight into the optimizer's flat buffer, NOTHING is computed here -- the arguments wait in link.pending until that
    layer's backward has written its data gradient dx, and ONE pass then does dx += w * (sum of the halves of g) [* ELU'(e)] and the
    weight's dot product (flush_pending): 2 g + e + dx read, dx written, where de written here + autograd's add moved 2 g + e read, de
    written, de + dx read, sum written."""
    if not want_e and not want_w:
        return None, None
    ds = rs = None
    if want_w:
        ds, rs = _grad_target(param)
    if (defer and SKIP_DEFER and want_e and link is not None and link.accumulates and rs is None and not torch.is_grad_enabled()):
        link.pending.append((g, e, weights, idx, reps, ds))
        return None, None
    de = new_cl16(*e.shape, e.device, e.dtype) if want_e else None
    gate = link is not None and link.gated
    check(lib16(e).tt_skip_join16_bwd(ptr(g), ptr(e), ptr(weights), idx, ptr(de), ptr(ds), e.numel(), reps, int(gate), stream_ptr()),
          'tt_skip_join16_bwd')
    return de, rs


# TTRAP_SKIP_RIDE=0 / ops.SKIP_RIDE = False: a parked skip-join backward is applied by a pass of its own (flush_pending) instead of riding on the
# gated epilogue of the encoder level's first block (A/B)
SKIP_RIDE = os.environ.get('TTRAP_SKIP_RIDE', '1') != '0'


def _riding_join(ctx, x0):
    """The one parked skip-join backward of ``ctx.link`` if it can ride on the level's gated first block (tt_wide_level_bwd_gated_join): the
    level gates, its first block has dilation 1, exactly one join is parked and its embedding IS the level's input; else None."""
    link = ctx.link
    pend = getattr(link, 'pending', None)
    if not SKIP_RIDE or not ctx.gate or not pend or len(pend) != 1 or ctx.dilations[0] != 1:
        return None
    g, e, weights, idx, reps, ds = pend[0]
    if e.data_ptr() != x0.data_ptr() or e.shape != x0.shape or g.dtype != x0.dtype or g.size(0) != reps * x0.size(0):
        return None
    return pend[0]


def flush_pending(link, dx):
    """Called by the backward of an encoder layer that consumes a linked tensor, after it has written the data gradient ``dx`` of that tensor
    (gated iff link.gated): fold the parked skip-join backwards into it (see _join_backward).  Inside the caller's loss_scaled scope."""
    if not getattr(link, 'pending', None):          # (None, nothing parked, or a link-like object of a stage-wise test)
        return
    pend, link.pending = link.pending, []
    for g, e, weights, idx, reps, ds in pend:
        if dx is None:                      # the layer's input wanted no gradient: only the weight's share is left to do
            if ds is not None:
                check(lib16(e).tt_skip_join16_bwd(ptr(g), ptr(e), ptr(weights), idx, None, ptr(ds), e.numel(), reps, 0, stream_ptr()), 'tt_skip_join16_bwd')
            continue
        flags = (1 if link.gated else 0) | 2
        check(lib16(e).tt_skip_join16_bwd(ptr(g), ptr(e), ptr(weights), idx, ptr(dx), ptr(ds), e.numel(), reps, flags, stream_ptr()), 'tt_skip_join16_bwd')


class SkipJoin16Fn(torch.autograd.Function):
    """out = y + weights[idx] * e on cl16 tensors in one pass (tt_skip_join16_fwd; reference modules.py:112 and :569-589).  y may hold
    ``reps`` = 2 batches back to back (TimbreTrap.decode_pair) that both take e.  Backward, one pass (tt_skip_join16_bwd): dy is the
    incoming gradient itself, de = weights[idx] * (sum over the halves), d weights[idx] += <sum over the halves, e>.  ``link``: e is the RAW
    output of a 16-bit strided layer whose backward may take its gradient already gated (GateLink, read at backward time): de then
    carries ELU'(e) -- what GateTapFn + tt_gate16 did in two more passes."""

    @staticmethod
    def forward(ctx, y, e, weights, idx, link, defer):
        B, C, H, T = e.shape
        reps = y.size(0) // B
        ctx.defer = defer
        if y.dtype != e.dtype or y.shape[1:] != e.shape[1:] or reps * B != y.size(0) or reps not in (1, 2):
            raise ValueError('skip join of %s %s and %s %s tensors' % (tuple(y.shape), y.dtype, tuple(e.shape), e.dtype))
        out = new_cl16(reps * B, C, H, T, y.device, y.dtype)
        check(lib16(y).tt_skip_join16_fwd(ptr(y), ptr(e), ptr(weights), idx, ptr(out), e.numel(), reps, stream_ptr()), 'tt_skip_join16_fwd')
        ctx.idx, ctx.reps, ctx.link, ctx.param = idx, reps, link, weights
        ctx.save_for_backward(e, weights)
        return out

    @staticmethod
    def backward(ctx, g):
        e, weights = ctx.saved_tensors
        B, C, H, T = e.shape
        g = _as_cl16(g, e.dtype)
        de, rs = _join_backward(g, e, weights, ctx.idx, ctx.reps, ctx.link, ctx.param, ctx.needs_input_grad[1], ctx.needs_input_grad[2], ctx.defer)
        return g, de, rs, None, None, None


def skip_join(y, skip):
    """y + skip for the decoder's joins: ``skip`` is a tensor (already scaled: apply_skip_connections) or a SkipJoin."""
    if not isinstance(skip, SkipJoin):
        return add(y, skip)
    e, w = skip.e, skip.weights
    if (is_cl16(y) and is_cl16(e) and y.dtype == e.dtype and e.numel() % 8 == 0 and w.dtype == torch.float32 and w.is_contiguous()
            and y.shape[1:] == e.shape[1:] and y.size(0) in (e.size(0), 2 * e.size(0))):
        return SkipJoin16Fn.apply(y, e, w, skip.idx, skip.link, skip.defer)
    # any other layout: the two-step form on whatever the tensors are (the tap carries the gate of a linked embedding)
    scaled = scale(gate_tap(e, skip.link), w, skip.idx)
    if y.size(0) == 2 * e.size(0):
        scaled = torch.cat((scaled, scaled), dim=0)
    return add(y, scaled)


def add(a, b):
    """a + b for the skip joins: on cl16 tensors when either side is one (Add16Fn), AddFn otherwise."""
    if is_cl16(a) or is_cl16(b):
        return Add16Fn.apply(to_cl16(a), to_cl16(b))
    return AddFn.apply(a, b)


def scale(e, weights, i):
    """weights[i] * e (TimbreTrap.apply_skip_connections)."""
    if is_cl16(e):
        return Scale16Fn.apply(e, weights, i)
    return ScaleFn.apply(e, weights, i)


# TTRAP_SKIP_FOLD=0 / ops.SKIP_FOLD = False: the join behind a DecoderBlock as a pass of its own (SkipJoin16Fn) instead of the epilogue of the
# level's last block (A/B)
SKIP_FOLD = os.environ.get('TTRAP_SKIP_FOLD', '1') != '0'


def residual_level(x, blocks, out_x3=False, link=None, join=None):
    """
    block3(block2(block1(x))) for the ResidualConv2dBlock modules ``blocks``.  With ops.cl16_mode() the level runs
    on cl16 tensors (Level16Fn) and RETURNS a cl16 tensor -- the next layer either has a bf16 kernel or converts with
    to_planar32; otherwise the per-block fp32 path.  ``out_x3``: the caller's next layer takes a split-operand tensor (is_x3):
    honoured only where the level itself runs on them (x3_inference()).  ``join`` (a SkipJoin): the result + the weighted skip -- in the
    epilogue of the level's last block where the level runs on cl16 tensors (Level16JoinFn), else skip_join() behind it.
    """
    if join is not None:
        C, T = x.size(1), x.size(-1)
        e = join.e
        if (SKIP_FOLD and cl16_mode() and is_cl16(x) and is_cl16(e) and e.dtype == x.dtype and e.shape[1:] == x.shape[1:] and x.size(0) % e.size(0) == 0
                and x.size(0) // e.size(0) in (1, 2) and e.numel() % 8 == 0 and C in WIDE_CHANNELS and FUSED_RESBLOCK and (C != 4 or T % 2 == 0)
                and _i32_ok(x) and join.weights.dtype == torch.float32 and join.weights.is_contiguous()
                and all(b.conv1[0].weight.shape == (C, C, 3, 3) and 1 <= b.dilation <= 3 for b in blocks)):
            params = []
            for b in blocks:
                params += [b.conv1[0].weight, b.conv1[0].bias, b.conv2[0].weight, b.conv2[0].bias]
            return Level16JoinFn.apply(x, tuple(b.dilation for b in blocks), link, e, join.weights, join.idx, join.link, join.defer, *params)
        return skip_join(residual_level(x, blocks, False, link), join)
    if is_x3(x):
        return x3_level(x, blocks, out_x3)
    C, T = x.size(1), x.size(-1)
    if (cl16_mode() and C in WIDE_CHANNELS and FUSED_RESBLOCK and (C != 4 or T % 2 == 0) and _i32_ok(x)
            and all(b.conv1[0].weight.shape == (C, C, 3, 3) and 1 <= b.dilation <= 3 for b in blocks)):
        params = []
        for b in blocks:
            params += [b.conv1[0].weight, b.conv1[0].bias, b.conv2[0].weight, b.conv2[0].bias]
        # link: only when x is the producing layer's own output tensor (to_cl16 is the identity on it)
        return Level16Fn.apply(to_cl16(x), tuple(b.dilation for b in blocks), link if is_cl16(x) else None, *params)
    x = to_planar32(x)
    if (x3_inference() and X3N_INFER and C in X3N_CHANNELS and x.is_cuda and _x3_blocks_ok(C, blocks)
            and _x3_size_ok(x.size(0), x.size(2), x.size(3))):
        y = x3n_level(x, blocks)
        if x3_chain() or x3_vouched() or x3_range_ok(y):          # same range rule as the wide levels below
            return y
        with x3_disabled():
            return residual_level(x, blocks)
    if x3_inference() and C in X3_CHANNELS and x.is_cuda and _x3_blocks_ok(C, blocks):
        y = x3_level(x, blocks, out_x3)
        # outside TimbreTrap._inference (which checks its final result once) a level that went fp32 -> split -> fp32 vouches for
        # its own range: beyond +-65504 the split form is non-finite and the level is repeated on the fp32 kernels
        if x3_chain() or x3_vouched() or is_x3(y) or x3_range_ok(y):
            return y
        with x3_disabled():
            return residual_level(x, blocks)
    for b in blocks:
        x = b(x)
    return x


# ---- fp32-class inference on split fp16 operands ("x3", csrc/conv_x3.hip) -------------------------------------------------------
# Without autocast and without grad (evaluate.py:94-95, transcribe() / reconstruct()) the wide levels do not need the hidden
# activations the fp32 backward reads, and the fp32 matrix instructions are what bounds tt_resblock_fwd: the level then runs on
# (hi, lo) fp16 pairs -- fp32-level results (tests/test_gpu_x3.py: 2e-6 of the tensor's scale against float64) at the 16-bit matrix
# rate.  TTRAP_X3_INFER=0 / ops.X3_INFER = False keeps the fp32 kernels.
X3_INFER = os.environ.get('TTRAP_X3_INFER', '1') != '0'
X3_CHANNELS = (16, 32)
X3N_CHANNELS = (4, 8)                 # the narrow levels: lane-per-pixel split-operand blocks (tt_x3n_level_fwd), fp32 planar in and out
X3N_INFER = os.environ.get('TTRAP_X3N_INFER', '1') != '0'        # 0: the narrow levels stay on the exact-fp32 kernels (A/B, bench.py)
X3_SHAPES = {}                        # event key -> (B, C, H, T) of the last instrumented call (bench.py's roofline_x3_fwd)
_X3_LOCAL = threading.local()


def x3_inference():
    return (X3_INFER and not getattr(_X3_LOCAL, 'off', False) and not torch.is_grad_enabled() and precision() == 'fp32'
            and wide_storage() == 'fp32')


class x3_disabled:
    """Inside this scope the calling thread's no-grad fp32 forwards stay on the fp32 kernels (the range fallback below)."""

    def __enter__(self):
        self.prev = getattr(_X3_LOCAL, 'off', False)
        _X3_LOCAL.off = True
        return self

    def __exit__(self, *exc):
        _X3_LOCAL.off = self.prev
        return False


def x3_range_ok(out):
    """
    The split representation holds |v| <= 65504 (hi = fp16(v)); beyond that -- activations or weights -- csrc/conv_x3.hip returns
    NON-FINITE values, never a finite wrong number, where the reference's fp32 evaluation stays finite.  Callers that took the
    split-operand path check their result with this (one read-only pass and one host sync per inference call) and, when it is not
    finite, repeat the computation inside ``x3_disabled()`` on the fp32 kernels: a genuinely non-finite result (NaN weights) comes
    out non-finite again, an out-of-range one finite -- the reference's answer either way (round-4 advisor finding).
    """
    return bool(torch.isfinite(out.float().sum()))


class x3_vouched_scope:
    """Inside this scope an OUTER caller checks the final result of the no-grad forward once (TimbreTrap._inference: its logits;
    TimbreTrap.chunked_inference: the cross-faded coefficients of all chunks -- ONE reduction and ONE host sync per transcribe() /
    reconstruct()), so the levels that went fp32 -> split -> fp32 do not each vouch for their own range with a reduction and a sync of
    their own (round-5 advisor finding: eight per forward of a skip-connection model).  A value beyond the split format's range comes out
    NaN (hi = inf, lo = inf - inf) and stays NaN through every later layer, so the final check sees it."""

    def __enter__(self):
        self.prev = getattr(_X3_LOCAL, 'vouched', False)
        _X3_LOCAL.vouched = True
        return self

    def __exit__(self, *exc):
        _X3_LOCAL.vouched = self.prev
        return False


def x3_vouched():
    return getattr(_X3_LOCAL, 'vouched', False)


def _x3_size_ok(B, H, T):
    """The launchers of csrc/conv_x3.hip count tiles in 32-bit integers (64-bit element offsets): B * H * T / 16 stays far below 2^31."""
    return B * H * T < 2 ** 34


class x3_chain_scope:
    """
    Inside this scope (TimbreTrap._inference when there are no skip connections: the encoder's embeddings are dropped) the layers
    between two split-operand levels hand x3 tensors to each other -- torch.float16 tensors of shape (B, H, T, 2, C), the layout of
    csrc/conv_x3.hip -- instead of fp32 planar ones: no pack / unpack passes, strided layers on tt_x3_sconv_fwd / tt_x3_tconv_fwd.
    Such tensors never leave the model: every consumer without an x3 kernel converts (to_planar32).
    """

    def __init__(self, enabled=True):
        self.enabled = enabled

    def __enter__(self):
        self.prev = getattr(_X3_LOCAL, 'chain', False)
        _X3_LOCAL.chain = bool(self.enabled)
        return self

    def __exit__(self, *exc):
        _X3_LOCAL.chain = self.prev
        return False


def x3_chain():
    return getattr(_X3_LOCAL, 'chain', False) and x3_inference()


def is_x3(t):
    return t.dtype == torch.float16 and t.dim() == 5 and t.size(3) == 2 and t.is_contiguous()


def from_x3(t):
    """x3 (B, H, T, 2, C) -> fp32 planar (B, C, H, T) (tt_x3_unpack: hi + lo 2^-11, exact)."""
    B, H, T, _, C = t.shape
    y = torch.empty((B, C, H, T), dtype=torch.float32, device=t.device)
    check(_hip.lib().tt_x3_unpack(ptr(t), ptr(y), B, C, H, T, stream_ptr()), 'tt_x3_unpack')
    return y


def _x3_blocks_ok(C, blocks):
    return (FUSED_RESBLOCK and 1 <= len(blocks) <= 8
            and all(b.conv1[0].weight.shape == (C, C, 3, 3) and b.conv2[0].weight.shape == (C, C, 1, 1) and 1 <= b.dilation <= 3
                    for b in blocks))


def x3_level(x, blocks, out_x3=False):
    """block_n(...block1(x)) without an autograd graph: tt_x3_level_fwd.  x: fp32 planar (B,C,H,T) or an x3 tensor; returns an x3
    tensor if out_x3 (the caller's next layer takes one), else fp32 planar."""
    in_x3 = is_x3(x)
    if in_x3:
        B, H, T, _, C = x.shape
        if C not in X3_CHANNELS or not _x3_blocks_ok(C, blocks) or not _x3_size_ok(B, H, T):
            with x3_disabled():
                return residual_level(from_x3(x), blocks)
    else:
        x = _f32c(x)
        B, C, H, T = x.shape
        if not _x3_size_ok(B, H, T):
            with x3_disabled():
                return residual_level(x, blocks)
    lib, st = _hip.lib(), stream_ptr()
    n = len(blocks)
    params = [[_f32c(t.detach()) for t in (b.conv1[0].weight, b.conv1[0].bias, b.conv2[0].weight, b.conv2[0].bias)] for b in blocks]
    _hip.require_cuda(x, params[0][0])
    arr = lambda j: (ctypes.c_void_p * n)(*[p[j].data_ptr() for p in params])
    ws = torch.empty(lib.tt_x3_level_scratch_bytes(B, C, H, T), dtype=torch.uint8, device=x.device)
    y = (torch.empty((B, H, T, 2, C), dtype=torch.float16, device=x.device) if out_x3
         else torch.empty((B, C, H, T), dtype=torch.float32, device=x.device))
    if _hip.EVENT_LOG is not None:
        # bench.py's instrumented steps: the same launches as tt_x3_level_fwd, one at a time, each block between its own pair of events
        half = ws.numel() // 2
        buf = (ws[:half], ws[half:])
        cur = x
        if not in_x3:
            check(lib.tt_x3_pack(ptr(x), ptr(buf[0]), B, C, H, T, st), 'tt_x3_pack')
            cur = buf[0]
        for i, (b, p) in enumerate(zip(blocks, params)):
            last = i == n - 1
            dst = y if last else (buf[1] if cur is buf[0] else buf[0])
            with _hip.timed('x3_rb_fwd_C%d' % C):
                check(lib.tt_x3_rb_fwd(ptr(cur), ptr(p[0]), ptr(p[1]), ptr(p[2]), ptr(p[3]), ptr(dst), int(last and not out_x3), B, C, H, T,
                                       b.dilation, st), 'tt_x3_rb_fwd')
            cur = dst
        X3_SHAPES['x3_rb_fwd_C%d' % C] = (B, C, H, T)
        return y
    check(lib.tt_x3_level_fwd(n, ptr(x), int(in_x3), ptr(y), int(out_x3), arr(0), arr(1), arr(2), arr(3),
                              (ctypes.c_int * n)(*[b.dilation for b in blocks]), ptr(ws), B, C, H, T, st), 'tt_x3_level_fwd')
    return y


def x3n_level(x, blocks):
    """block_n(...block1(x)) of a NARROW level (C = 4, 8) without an autograd graph: tt_x3n_level_fwd, fp32 planar (B,C,H,T) in and
    out, split-operand tensors between the blocks."""
    x = _f32c(x)
    B, C, H, T = x.shape
    lib, st = _hip.lib(), stream_ptr()
    n = len(blocks)
    params = [[_f32c(t.detach()) for t in (b.conv1[0].weight, b.conv1[0].bias, b.conv2[0].weight, b.conv2[0].bias)] for b in blocks]
    _hip.require_cuda(x, params[0][0])
    arr = lambda j: (ctypes.c_void_p * n)(*[p[j].data_ptr() for p in params])
    ws = torch.empty(lib.tt_x3n_level_scratch_bytes(B, C, H, T), dtype=torch.uint8, device=x.device)
    y = torch.empty((B, C, H, T), dtype=torch.float32, device=x.device)
    if _hip.EVENT_LOG is not None:
        # bench.py's instrumented steps: the same launches, each block between its own pair of events
        half = ws.numel() // 2
        buf = (ws[:half], ws[half:])
        cur = x
        for i, (b, p) in enumerate(zip(blocks, params)):
            last = i == n - 1
            dst = y if last else buf[i & 1]
            with _hip.timed('x3n_rb_fwd_C%d' % C):
                check(lib.tt_x3n_rb_fwd(ptr(cur), int(i == 0), ptr(p[0]), ptr(p[1]), ptr(p[2]), ptr(p[3]), ptr(dst), int(last), B, C, H, T,
                                        b.dilation, st), 'tt_x3n_rb_fwd')
            cur = dst
        X3_SHAPES['x3n_rb_fwd_C%d' % C] = (B, C, H, T)
        return y
    check(lib.tt_x3n_level_fwd(n, ptr(x), ptr(y), arr(0), arr(1), arr(2), arr(3), (ctypes.c_int * n)(*[b.dilation for b in blocks]), ptr(ws),
                               B, C, H, T, st), 'tt_x3n_level_fwd')
    return y


def x3_strided_conv(x, w, b, out_x3):
    """EncoderBlock.sconv with split operands (tt_x3_sconv_fwd).  x: an x3 tensor (C = 16, 32) or fp32 planar (C = 8: the layer that
    enters the split-operand part); returns x3 (B, Hout, T, 2, 2C) or fp32 planar (B, 2C, Hout, T)."""
    pin = not is_x3(x)
    if pin:
        x = _f32c(x)
        B, C, H, T = x.shape
    else:
        B, H, T, _, C = x.shape
    Ho = (H - 4) // 2 + 1
    w, b = _f32c(w.detach()), _f32c(b.detach())
    _hip.require_cuda(x, w)
    y = (torch.empty((B, Ho, T, 2, 2 * C), dtype=torch.float16, device=x.device) if out_x3
         else torch.empty((B, 2 * C, Ho, T), dtype=torch.float32, device=x.device))
    with _hip.timed('x3_sconv_C%d' % C):
        check(_hip.lib().tt_x3_sconv_fwd(ptr(x), int(pin), ptr(w), ptr(b), ptr(y), int(not out_x3), B, C, H, T, stream_ptr()),
              'tt_x3_sconv_fwd')
    return y


def x3_transposed_conv(x, w, b, out_pad, out_x3):
    """DecoderBlock.tconv with split operands (tt_x3_tconv_fwd).  x: an x3 tensor with 2C = 32 channels or fp32 planar with 2C = 64;
    returns x3 (B, Hout, T, 2, C) or fp32 planar (B, C, Hout, T)."""
    pin = not is_x3(x)
    if pin:
        x = _f32c(x)
        B, C2, H, T = x.shape
    else:
        B, H, T, _, C2 = x.shape
    C, Ho = C2 // 2, 2 * H + 2 + out_pad
    w, b = _f32c(w.detach()), _f32c(b.detach())
    _hip.require_cuda(x, w)
    y = (torch.empty((B, Ho, T, 2, C), dtype=torch.float16, device=x.device) if out_x3
         else torch.empty((B, C, Ho, T), dtype=torch.float32, device=x.device))
    with _hip.timed('x3_tconv_C%d' % C):
        check(_hip.lib().tt_x3_tconv_fwd(ptr(x), int(pin), ptr(w), ptr(b), ptr(y), int(not out_x3), B, C, H, T, out_pad, stream_ptr()),
              'tt_x3_tconv_fwd')
    return y


class LatentEncodeFn(torch.autograd.Function):
    """Encoder.convlat: Conv2d(C, D, (E,1)) collapsing the frequency axis = per-clip GEMM (D x C*E)(C*E x T)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x, w = _f32c(x), _f32c(w)
        B, C, E, T = x.shape
        D, K = w.size(0), C * E
        y = torch.empty((B, D, T), dtype=torch.float32, device=x.device)
        check(_hip.lib().tt_gemm(ptr(w), ptr(x), ptr(y), ptr(b), D, T, K, 0, 0, K, T, T, B, 0, K * T, D * T, 0,
                                 1.0, 0.0, 1 if b is not None else 0, 1, ACT_NONE, stream_ptr()), 'tt_gemm(convlat)')
        ctx.has_bias = b is not None
        ctx.params = (w, b)
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _f32c(dy)
        B, C, E, T = x.shape
        D, K = w.size(0), C * E
        lib, st = _hip.lib(), stream_ptr()
        dx = dw = db = rw = rb = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            check(lib.tt_gemm(ptr(w), ptr(dy), ptr(dx), None, K, T, D, 1, 0, K, T, T, B, 0, D * T, K * T, 0,
                              1.0, 0.0, 0, 1, ACT_NONE, st), 'tt_gemm(convlat dgrad)')
        if ctx.needs_input_grad[1]:
            dw, rw = _grad_target(ctx.params[0])
            check(lib.tt_gemm(ptr(dy), ptr(x), ptr(dw), None, D, K, T, 0, 1, T, T, K, B, D * T, K * T, 0, 1,
                              1.0, 1.0, 0, 1, ACT_NONE, st), 'tt_gemm(convlat wgrad)')
            if ctx.has_bias:
                db, rb = _grad_target(ctx.params[1])
                check(lib.tt_channel_sum(ptr(dy), ptr(db), B, D, T, st), 'tt_channel_sum')
        return dx, rw, rb


class LatentDecodeFn(torch.autograd.Function):
    """Decoder.convin: ConvTranspose2d(D+1, C, (E,1)) + ELU = per-clip GEMM (C*E x D+1)(D+1 x T)."""

    @staticmethod
    def forward(ctx, z, w, b):
        z, w = _f32c(z), _f32c(w)
        B, K, T = z.shape
        C, E = w.size(1), w.size(2)
        Mo = C * E
        y = torch.empty((B, C, E, T), dtype=torch.float32, device=z.device)
        check(_hip.lib().tt_gemm(ptr(w), ptr(z), ptr(y), ptr(b), Mo, T, K, 1, 0, Mo, T, T, B, 0, K * T, Mo * T, 0,
                                 1.0, 0.0, 2 if b is not None else 0, E, ACT_ELU, stream_ptr()), 'tt_gemm(dec convin)')
        ctx.has_bias = b is not None
        ctx.params = (w, b)
        ctx.save_for_backward(z, w, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        z, w, y = ctx.saved_tensors
        dy = _f32c(dy)
        B, K, T = z.shape
        C, E = w.size(1), w.size(2)
        Mo = C * E
        lib, st = _hip.lib(), stream_ptr()
        g = torch.empty_like(dy)
        check(lib.tt_elu_bwd(ptr(dy), ptr(y), ptr(g), dy.numel(), st), 'tt_elu_bwd')
        dz = dw = db = rw = rb = None
        if ctx.needs_input_grad[0]:
            dz = torch.empty_like(z)
            check(lib.tt_gemm(ptr(w), ptr(g), ptr(dz), None, K, T, Mo, 0, 0, Mo, T, T, B, 0, Mo * T, K * T, 0,
                              1.0, 0.0, 0, 1, ACT_NONE, st), 'tt_gemm(dec convin dgrad)')
        if ctx.needs_input_grad[1]:
            dw, rw = _grad_target(ctx.params[0])
            check(lib.tt_gemm(ptr(z), ptr(g), ptr(dw), None, K, Mo, T, 0, 1, T, T, Mo, B, K * T, Mo * T, 0, 1,
                              1.0, 1.0, 0, 1, ACT_NONE, st), 'tt_gemm(dec convin wgrad)')
            if ctx.has_bias:
                db, rb = _grad_target(ctx.params[1])
                check(lib.tt_channel_sum(ptr(g), ptr(db), B, C, E * T, st), 'tt_channel_sum')
        return dz, rw, rb


def _lat16_ok(CT, D, E, T, b):
    return b is not None and T % 16 == 0 and ((CT == 32 and D <= 48) or (CT == 64 and D <= 144))


class LatEnc16Fn(torch.autograd.Function):
    """Encoder.convlat on the cl16 top embedding (csrc/latent_bf16.hip): (B,CT,E,T) cl16 -> latents (B,D,T) fp32."""

    @staticmethod
    def forward(ctx, x, w, b, link=None):
        B, CT, E, T = x.shape
        D = w.size(0)
        lib = lib16(x)
        y = torch.empty((B, D, T), dtype=torch.float32, device=x.device)
        ws = torch.empty(lib.tt_latent16_scratch_bytes(B, CT, D, E, T), dtype=torch.uint8, device=x.device)
        check(lib.tt_latent16_contract(ptr(x), None, ptr(w), ptr(b), ptr(y), ptr(ws), B, CT, D, D, E, T, stream_ptr()),
              'tt_latent16_contract')
        # x is the output of the encoder's last strided layer (+ ELU): its gradient goes back gated (GateLink)
        ctx.gate = bool(link is not None and link.producer and ctx.needs_input_grad[0])
        if ctx.gate:
            link.gated = True
        ctx.link = link
        if link is not None and ctx.needs_input_grad[0]:
            link.accumulates = True         # flush_pending in backward
        ctx.params = (w, b)
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        B, CT, E, T = x.shape
        D = w.size(0)
        lib, st = lib16(x), stream_ptr()
        dy = _f32c(dy)
        ws = torch.empty(lib.tt_latent16_scratch_bytes(B, CT, D, E, T), dtype=torch.uint8, device=x.device)
        dx = rw = rb = None
        if ctx.needs_input_grad[0]:
            dx = new_cl16(B, CT, E, T, x.device, x.dtype)
            if ctx.gate:
                check(lib.tt_latent16_expand_gated(ptr(dy), ptr(w), ptr(x), ptr(dx), ptr(ws), B, CT, D, E, T, st), 'tt_latent16_expand_gated')
            else:
                check(lib.tt_latent16_expand(ptr(dy), D, 0.0, ptr(w), None, ptr(dx), ptr(ws), B, CT, D, E, T, st), 'tt_latent16_expand')
        flush_pending(ctx.link, dx)
        if ctx.needs_input_grad[1]:
            dw, rw = _grad_target(ctx.params[0])
            check(lib.tt_latent16_wgrad(ptr(dy), D, 0.0, ptr(x), None, ptr(dw), None, ptr(ws), B, CT, D, E, T, st), 'tt_latent16_wgrad')
            db, rb = _grad_target(ctx.params[1])
            check(lib.tt_channel_sum(ptr(dy), ptr(db), B, D, T, st), 'tt_channel_sum')
        return dx, rw, rb, None


class LatDec16Fn(torch.autograd.Function):
    """
    Decoder.convin producing the cl16 top embedding: z (B,Dz,T) fp32 -> ELU(tconv) (B,CT,E,T) cl16.  ``fill`` is None (z carries
    all D = w.size(0) input channels) or the value of a constant LAST channel that z does not carry (Dz = D - 1): the
    transcription switch of TimbreTrap.decode (reference modules.py:139-142) without building the concatenated tensor.
    """

    @staticmethod
    def forward(ctx, z, w, b, fill, link=None):
        z = _f32c(z)
        B, Dz, T = z.shape
        D, CT, E = w.size(0), w.size(1), w.size(2)
        # the pregated backward carries the bias gradient in a free input row of the weight gradient: the library says where it can
        ctx.link = link if (link is not None and lib16(cl16_dtype()).tt_latent16_pregated_ok(CT, D)) else None
        if ctx.link is not None:
            ctx.link.producer = True
        y = new_cl16(B, CT, E, T, z.device, cl16_dtype())
        lib = lib16(y)
        ws = torch.empty(lib.tt_latent16_scratch_bytes(B, CT, D, E, T), dtype=torch.uint8, device=z.device)
        ctx.fill = 0.0 if fill is None else float(fill)
        check(lib.tt_latent16_expand(ptr(z), Dz, ctx.fill, ptr(w), ptr(b), ptr(y), ptr(ws), B, CT, D, E, T, stream_ptr()),
              'tt_latent16_expand')
        ctx.params = (w, b)
        ctx.save_for_backward(z, w, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        z, w, y = ctx.saved_tensors
        B, Dz, T = z.shape
        D, CT, E = w.size(0), w.size(1), w.size(2)
        lib, st = lib16(y), stream_ptr()
        g = _as_cl16(dy, y.dtype)
        ws = torch.empty(lib.tt_latent16_scratch_bytes(B, CT, D, E, T), dtype=torch.uint8, device=z.device)
        dz = rw = rb = None
        pre = ctx.link is not None and ctx.link.gated           # the transposed layer behind left dy * ELU'(y)
        if ctx.needs_input_grad[0]:
            dz = torch.empty_like(z)
            if pre:
                check(lib.tt_latent16_contract_pregated(ptr(g), ptr(w), ptr(dz), ptr(ws), B, CT, D, Dz, E, T, st), 'tt_latent16_contract_pregated')
            else:
                check(lib.tt_latent16_contract(ptr(g), ptr(y), ptr(w), None, ptr(dz), ptr(ws), B, CT, D, Dz, E, T, st), 'tt_latent16_contract')
        if ctx.needs_input_grad[1]:
            dw, rw = _grad_target(ctx.params[0])
            db, rb = _grad_target(ctx.params[1])
            if pre:
                check(lib.tt_latent16_wgrad_pregated(ptr(z), Dz, ctx.fill, ptr(g), ptr(dw), ptr(db), ptr(ws), B, CT, D, E, T, st),
                      'tt_latent16_wgrad_pregated')
            else:
                check(lib.tt_latent16_wgrad(ptr(z), Dz, ctx.fill, ptr(g), ptr(y), ptr(dw), ptr(db), ptr(ws), B, CT, D, E, T, st), 'tt_latent16_wgrad')
        return dz, rw, rb, None, None


X3_LATENT_SHAPES = ((64, 128), (32, 32))          # (channels of the top embedding, latent size) with split-operand latent heads


def x3_latent_ok(C, D, w_enc=None, w_dec=None):
    return ((C, D) in X3_LATENT_SHAPES and (w_enc is None or (w_enc.dim() == 4 and w_enc.shape[:2] == (D, C) and w_enc.size(3) == 1))
            and (w_dec is None or (w_dec.dim() == 4 and w_dec.shape[:2] == (D + 1, C) and w_dec.size(3) == 1)))


def x3_latent_encode(top, w, b):
    """Encoder.convlat on an x3 embedding (B, E, T, 2, C) -> latents (B, D, T) fp32 (tt_x3_latent_encode)."""
    B, E, T, _, C = top.shape
    D = w.size(0)
    lib = _hip.lib()
    w = _f32c(w.detach())
    b = None if b is None else _f32c(b.detach())
    ws = torch.empty(lib.tt_x3_latent_scratch_bytes(C, E, D), dtype=torch.uint8, device=top.device)
    z = torch.empty((B, D, T), dtype=torch.float32, device=top.device)
    with _hip.timed('x3_latent_encode'):
        check(lib.tt_x3_latent_encode(ptr(top), ptr(w), ptr(b), ptr(z), ptr(ws), B, C, E, D, T, stream_ptr()), 'tt_x3_latent_encode')
    return z


def x3_latent_decode(z, w, b, fill, out_x3):
    """Decoder.convin + ELU with split operands: latents (B, D or D + 1, T) fp32 -> x3 (B, E, T, 2, C) or fp32 planar (tt_x3_latent_decode)."""
    z = _f32c(z)
    B, Dz, T = z.shape
    D, C, E = w.size(0) - 1, w.size(1), w.size(2)
    lib = _hip.lib()
    w = _f32c(w.detach())
    b = None if b is None else _f32c(b.detach())
    ws = torch.empty(lib.tt_x3_latent_scratch_bytes(C, E, D), dtype=torch.uint8, device=z.device)
    y = (torch.empty((B, E, T, 2, C), dtype=torch.float16, device=z.device) if out_x3
         else torch.empty((B, C, E, T), dtype=torch.float32, device=z.device))
    with _hip.timed('x3_latent_decode'):
        check(lib.tt_x3_latent_decode(ptr(z), Dz, float(fill) if fill is not None else 0.0, ptr(w), ptr(b), ptr(y), int(not out_x3), ptr(ws),
                                      B, C, E, D, T, stream_ptr()), 'tt_x3_latent_decode')
    return y


def latent_encode(top, w, b, link=None):
    """Encoder.convlat (modules.py:446)."""
    if is_x3(top):
        if x3_latent_ok(top.size(4), w.size(0), w_enc=w) and w.size(2) == top.size(1):
            return x3_latent_encode(top, w, b)
        top = from_x3(top)
    if is_cl16(top) and _f32ok(w) and _lat16_ok(top.size(1), w.size(0), top.size(2), top.size(3), b) and w.shape[1:] == (top.size(1), top.size(2), 1):
        return LatEnc16Fn.apply(top, w, b, link)
    return LatentEncodeFn.apply(to_planar32(top), w, b)


def latent_decode(z, w, b, fill=None, out_x3=False, link=None):
    """
    Decoder.convin (modules.py:534) + ELU; cl16 output in the bf16 mode.  ``fill``: value of a constant last input channel that z
    does not carry (see LatDec16Fn); on the fp32 path the channel is concatenated like the reference does.  ``out_x3`` (inside
    ops.x3_chain_scope): the next layer takes a split-operand tensor.
    """
    Dz = z.size(1) + (fill is not None)
    if (out_x3 and x3_chain() and z.dim() == 3 and z.is_cuda and w.size(0) == Dz and x3_latent_ok(w.size(1), Dz - 1, w_dec=w)
            and w.size(2) * w.size(1) * 4 + 2 * ((Dz - 1) // 32) * (w.size(1) // 16) * 2048 <= 160 * 1024):
        return x3_latent_decode(z, w, b, fill, True)
    if (cl16_mode() and FUSED_RESBLOCK and _f32ok(w) and z.dim() == 3 and w.size(0) == Dz and w.size(3) == 1
            and _lat16_ok(w.size(1), w.size(0), w.size(2), z.size(2), b)):
        return LatDec16Fn.apply(z, w, b, fill, link)
    if fill is not None:
        z = torch.cat((z, torch.full_like(z[..., :1, :], float(fill))), dim=-2)
    return LatentDecodeFn.apply(z, w, b)


def _f32ok(w):
    return w.dtype == torch.float32 and w.is_contiguous()


# ---- objectives ------------------------------------------------------------------------------------

def _partials(device):
    return torch.empty(1024, dtype=torch.float64, device=device)


# TTRAP_LOSS_FUSED=0 / ops.LOSS_FUSED = False: the squared-error losses compute their gradients in backward from the saved operands (A/B)
LOSS_FUSED = os.environ.get('TTRAP_LOSS_FUSED', '1') != '0'


class SqDiffLossFn(torch.autograd.Function):
    """loss = scale * sum((a - b)^2); gradients flow into both arguments.  With grad enabled the forward pass, which reads both operands
    anyway, also writes the gradient for an incoming scalar of 1 (tt_sqdiff_sum_grad); backward is a launch that checks the incoming scalar
    on the device and multiplies only if it is not 1 (tt_sqdiff_rescale) -- no second read of the operands.  A second backward through the
    same node (retain_graph) recomputes from the operands (tt_sqdiff_bwd)."""

    @staticmethod
    def forward(ctx, a, b, scale):
        _hip.require_cuda(a, b)
        a, b = _f32c(a), _f32c(b)
        loss = torch.empty((), dtype=torch.float32, device=a.device)
        need = (ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        ctx.grads = None
        if LOSS_FUSED and any(need) and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0:
            da = torch.empty_like(a) if need[0] else None
            db = torch.empty_like(b) if need[1] else None
            check(_hip.lib().tt_sqdiff_sum_grad(ptr(a), ptr(b), ptr(loss), ptr(_partials(a.device)), a.numel(), scale, ptr(da), ptr(db),
                                                stream_ptr()), 'tt_sqdiff_sum_grad')
            ctx.grads = (da, db)
        else:
            check(_hip.lib().tt_sqdiff_sum(ptr(a), ptr(b), ptr(loss), ptr(_partials(a.device)), a.numel(), scale,
                                           stream_ptr()), 'tt_sqdiff_sum')
        ctx.scale = scale
        ctx.save_for_backward(a, b)
        return loss

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = _f32c(g)
        if ctx.grads is not None:
            da, db = ctx.grads
            ctx.grads = None                                     # single use: autograd may accumulate into these buffers in place
            check(_hip.lib().tt_sqdiff_rescale(ptr(da), ptr(db), ptr(g), a.numel(), stream_ptr()), 'tt_sqdiff_rescale')
            return da, db, None
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        if da is not None or db is not None:
            check(_hip.lib().tt_sqdiff_bwd(ptr(a), ptr(b), ptr(g), ctx.scale, ptr(da), ptr(db), a.numel(),
                                           stream_ptr()), 'tt_sqdiff_bwd')
        return da, db, None


class SqDiff2Fn(torch.autograd.Function):
    """
    (scale * sum((a1 - b)^2), scale * sum((a2 - b)^2)): the two consistency terms (reference objectives.py:77-104), which share their
    non-detached second operand.  Same sums as two SqDiffLossFn; the backward is ONE pass (tt_sqdiff2_bwd) that writes the gradient
    of b once, as the sum of both terms -- two SqDiffLossFn left that sum to autograd (an extra elementwise pass over the logits).
    """

    @staticmethod
    def forward(ctx, a1, a2, b, scale):
        _hip.require_cuda(a1, b)
        # tt_sqdiff2_bwd reads 16 bytes per lane: a contiguous view at an odd storage offset (a batch slice ``x[k:]`` whose offset is
        # not a multiple of four floats) is copied once here rather than failing in backward after the forward has succeeded
        a1, a2, b = (t if t.data_ptr() % 16 == 0 else t.clone() for t in (_f32c(a1), _f32c(a2), _f32c(b)))
        lib, st = _hip.lib(), stream_ptr()
        l1 = torch.empty((), dtype=torch.float32, device=b.device)
        l2 = torch.empty((), dtype=torch.float32, device=b.device)
        ctx.grads = None
        if LOSS_FUSED and any(ctx.needs_input_grad[:3]):
            # one pass over a1, a2, b for both sums and the three gradients at unit incoming scale (tt_sqdiff2_sum_grad; see SqDiffLossFn)
            da1, da2 = torch.empty_like(a1), torch.empty_like(a2)
            db = torch.empty_like(b) if ctx.needs_input_grad[2] else None
            check(lib.tt_sqdiff2_sum_grad(ptr(a1), ptr(a2), ptr(b), ptr(l1), ptr(l2), ptr(torch.empty(2048, dtype=torch.float64, device=b.device)),
                                          b.numel(), scale, ptr(da1), ptr(da2), ptr(db), st), 'tt_sqdiff2_sum_grad')
            ctx.grads = (da1, da2, db)
        else:
            check(lib.tt_sqdiff_sum(ptr(a1), ptr(b), ptr(l1), ptr(_partials(b.device)), b.numel(), scale, st), 'tt_sqdiff_sum')
            check(lib.tt_sqdiff_sum(ptr(a2), ptr(b), ptr(l2), ptr(_partials(b.device)), b.numel(), scale, st), 'tt_sqdiff_sum')
        ctx.scale = scale
        ctx.save_for_backward(a1, a2, b)
        return l1, l2

    @staticmethod
    def backward(ctx, g1, g2):
        a1, a2, b = ctx.saved_tensors
        g1 = None if g1 is None else _f32c(g1)
        g2 = None if g2 is None else _f32c(g2)
        if ctx.grads is not None:
            da1, da2, db = ctx.grads
            ctx.grads = None
            check(_hip.lib().tt_sqdiff2_rescale(ptr(da1), ptr(da2), ptr(db), ptr(g1), ptr(g2), b.numel(), stream_ptr()), 'tt_sqdiff2_rescale')
            return (da1 if ctx.needs_input_grad[0] else None, da2 if ctx.needs_input_grad[1] else None, db, None)
        da1 = torch.empty_like(a1) if ctx.needs_input_grad[0] else None
        da2 = torch.empty_like(a2) if ctx.needs_input_grad[1] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[2] else None
        if da1 is not None or da2 is not None or db is not None:
            check(_hip.lib().tt_sqdiff2_bwd(ptr(a1), ptr(a2), ptr(b), ptr(g1), ptr(g2), ctx.scale, ptr(da1), ptr(da2), ptr(db), b.numel(),
                                            stream_ptr()), 'tt_sqdiff2_bwd')
        return da1, da2, db, None


class ActivationsFn(torch.autograd.Function):
    """tanh(|re + i im|) over the channel pair (TimbreTrap.to_activations, modules.py:287)."""

    @staticmethod
    def forward(ctx, coefficients):
        _hip.require_cuda(coefficients)
        c = _f32c(coefficients)
        B, _, F, T = c.shape
        act = torch.empty((B, F, T), dtype=torch.float32, device=c.device)
        check(_hip.lib().tt_activations_fwd(ptr(c), ptr(act), B, F, T, stream_ptr()), 'tt_activations_fwd')
        ctx.save_for_backward(c, act)
        return act

    @staticmethod
    def backward(ctx, dact):
        c, act = ctx.saved_tensors
        B, _, F, T = c.shape
        dc = torch.empty_like(c)
        check(_hip.lib().tt_activations_bwd(ptr(c), ptr(act), ptr(_f32c(dact)), ptr(dc), B, F, T, stream_ptr()),
              'tt_activations_bwd')
        return dc


class TranscriptionLossFn(torch.autograd.Function):
    """compute_transcription_loss (objectives.py:36-74); gradient w.r.t. the estimate only."""

    @staticmethod
    def forward(ctx, estimate, target, weighted):
        _hip.require_cuda(estimate, target)
        e, t = _f32c(estimate), _f32c(target)
        B, F, T = e.shape
        loss = torch.empty((), dtype=torch.float32, device=e.device)
        fs = torch.empty((B, T), dtype=torch.float32, device=e.device) if weighted else None
        ctx.grad = None
        if LOSS_FUSED and ctx.needs_input_grad[0]:               # the gradient for an incoming scalar of 1 in the same pass (see SqDiffLossFn)
            de = torch.empty_like(e)
            check(_hip.lib().tt_transcription_loss_fwd_grad(ptr(e), ptr(t), ptr(loss), ptr(fs), ptr(_partials(e.device)), ptr(de),
                                                            B, F, T, int(weighted), stream_ptr()), 'tt_transcription_loss_fwd_grad')
            ctx.grad = de
        else:
            check(_hip.lib().tt_transcription_loss_fwd(ptr(e), ptr(t), ptr(loss), ptr(fs), ptr(_partials(e.device)),
                                                       B, F, T, int(weighted), stream_ptr()), 'tt_transcription_loss_fwd')
        ctx.weighted = weighted
        ctx.save_for_backward(e, t, fs)
        return loss

    @staticmethod
    def backward(ctx, g):
        e, t, fs = ctx.saved_tensors
        B, F, T = e.shape
        if ctx.grad is not None:
            de, ctx.grad = ctx.grad, None
            check(_hip.lib().tt_sqdiff_rescale(ptr(de), None, ptr(_f32c(g)), de.numel(), stream_ptr()), 'tt_sqdiff_rescale')
            return de, None, None
        de = torch.empty_like(e)
        check(_hip.lib().tt_transcription_loss_bwd(ptr(e), ptr(t), ptr(fs), ptr(_f32c(g)), ptr(de), B, F, T,
                                                   int(ctx.weighted), stream_ptr()), 'tt_transcription_loss_bwd')
        return de, None, None


# ---- loss-scaled backward of the 16-bit layers (FP16_LOSS_SCALE above) -----------------------------------------------------------

def _scale_backward(cls, dtype_of):
    """Run cls.backward inside ``loss_scaled(<element type of the layer's 16-bit tensors>)``: its kernels then treat incoming 16-bit
    gradients as scaled and unscale every fp32 result (ConvIn16Fn / ConvOut16Fn do it around their single call)."""
    bwd = cls.backward

    def backward(ctx, *grads):
        with loss_scaled(dtype_of(ctx)):
            return bwd(ctx, *grads)
    cls.backward = staticmethod(backward)


_scale_backward(Level16Fn, lambda ctx: ctx.saved_tensors[-1].dtype)
_scale_backward(Level16JoinFn, lambda ctx: ctx.saved_tensors[-1].dtype)
_scale_backward(WideLevelFn, lambda ctx: ctx.dtype)
_scale_backward(SConv16Fn, lambda ctx: ctx.saved_tensors[0].dtype)
_scale_backward(TConv16Fn, lambda ctx: ctx.saved_tensors[0].dtype)
_scale_backward(LatEnc16Fn, lambda ctx: ctx.saved_tensors[0].dtype)
_scale_backward(LatDec16Fn, lambda ctx: ctx.saved_tensors[2].dtype)
_scale_backward(SkipJoin16Fn, lambda ctx: ctx.saved_tensors[0].dtype)


# ---- instrumentation (bench.py): bracket every forward / backward of the Functions above with HIP events ------------------------

def _instrument(cls, name, keyfn):
    """Wrap cls.forward / cls.backward so that, when _hip.EVENT_LOG is a dict, each call is bracketed by an event pair
    recorded on the launch stream under '<name>_fwd|bwd_<shape tag>' (no cost when logging is off)."""
    fwd, bwd = cls.forward, cls.backward

    def forward(ctx, *args):
        if _hip.EVENT_LOG is None:
            return fwd(ctx, *args)
        ctx._tt_key = keyfn(*args)
        ctx._tt_clips = args[0].size(0) if (torch.is_tensor(args[0]) and args[0].dim() >= 3) else None
        with _hip.timed('%s_fwd_%s' % (name, ctx._tt_key), clips=ctx._tt_clips):
            return fwd(ctx, *args)

    def backward(ctx, *grads):
        if _hip.EVENT_LOG is None:
            return bwd(ctx, *grads)
        with _hip.timed('%s_bwd_%s' % (name, getattr(ctx, '_tt_key', '?')), clips=getattr(ctx, '_tt_clips', None)):
            return bwd(ctx, *grads)
    cls.forward = staticmethod(forward)
    cls.backward = staticmethod(backward)


_instrument(ConvFn, 'conv', lambda x, w, b, cfg: '%dto%d' % ((x.size(1), w.size(0)) if cfg.kind == 'conv' else (x.size(1), w.size(1))))
_instrument(ResBlockFn, 'rb', lambda x, *a: 'C%d' % x.size(1))
_instrument(WideLevelFn, 'widelevel', lambda x, *a: 'C%d' % x.size(1))
_instrument(ConvIn16Fn, 'edge16', lambda x, *a: 'in')
_instrument(ConvOut16Fn, 'edge16', lambda x, *a: 'out')
_instrument(ConvOut16PairFn, 'edge16', lambda x, *a: 'out')
_instrument(Level16Fn, 'widelevel', lambda x, *a: 'C%d' % x.size(1))
_instrument(Level16JoinFn, 'widelevel', lambda x, *a: 'C%d' % x.size(1))
_instrument(SConv16Fn, 'sconv16', lambda x, *a: 'C%d' % x.size(1))
_instrument(TConv16Fn, 'tconv16', lambda x, w, *a: 'C%d' % w.size(1))
_instrument(ToCL16Fn, 'tocl16', lambda x: 'C%d' % x.size(1))
_instrument(ToPlanar32Fn, 'toplanar', lambda x: 'C%d' % x.size(1))
_instrument(StridedConvFn, 'sconv', lambda x, *a: 'C%d' % x.size(1))
_instrument(TransposedConvFn, 'tconv', lambda x, w, *a: 'C%d' % w.size(1))
_instrument(SkipJoin16Fn, 'skipjoin16', lambda y, e, *a: 'C%d' % e.size(1))
_instrument(LatEnc16Fn, 'latenc16', lambda x, *a: 'C%d' % x.size(1))
_instrument(LatDec16Fn, 'latdec16', lambda z, w, *a: 'C%d' % w.size(1))
_instrument(LatentEncodeFn, 'latenc', lambda x, *a: 'C%d' % x.size(1))
_instrument(LatentDecodeFn, 'latdec', lambda z, w, *a: 'C%d' % w.size(1))
_instrument(SqDiffLossFn, 'sqdiff', lambda a, *r: 'n')
_instrument(SqDiff2Fn, 'sqdiff2', lambda a, *r: 'n')
_instrument(ActivationsFn, 'act', lambda c: 'n')
_instrument(TranscriptionLossFn, 'trn', lambda e, *r: 'n')
