"""
GPU parity of the convolution kernels (general direct conv, fused ResidualConv2dBlock on fp32 MFMA,
latent GEMM) against torch-CPU float64 restatements and the golden fixtures from the reference.
"""

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import autoencoder as oae

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


@pytest.mark.parametrize('C,d', [(4, 1), (4, 3), (8, 2), (16, 1), (16, 3), (32, 1), (32, 2), (32, 3)])
@pytest.mark.parametrize('shape', [(2, 13, 70), (1, 9, 130), (1, 31, 64), (2, 20, 192), (1, 70, 256), (9, 5, 68)])
@pytest.mark.parametrize('save_hidden', [True, False])
def test_fused_resblock_forward_backward(C, d, shape, save_hidden, monkeypatch):
    from timbre_trap.framework import ops
    monkeypatch.setattr(ops, 'SAVE_HIDDEN', save_hidden)      # hidden activation kept vs recomputed in backward
    B, H, T = shape
    x = _rand(B, C, H, T, seed=1)
    w1 = _rand(C, C, 3, 3, seed=2, scale=1.0 / (3 * C ** 0.5))
    b1 = _rand(C, seed=3, scale=0.3)
    w2 = _rand(C, C, 1, 1, seed=4, scale=1.0 / C ** 0.5)
    b2 = _rand(C, seed=5, scale=0.3)
    gy = _rand(B, C, H, T, seed=6)

    ref_in = [t.double().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    sd = {'p.conv1.0.weight': ref_in[1], 'p.conv1.0.bias': ref_in[2], 'p.conv2.0.weight': ref_in[3], 'p.conv2.0.bias': ref_in[4]}
    yr = oae.residual_block(ref_in[0], sd, 'p', d)
    yr.backward(gy.double())

    dev = [t.cuda().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    assert ops.FUSED_RESBLOCK
    y = ops.residual_block(*dev, d)
    y.backward(gy.cuda())
    assert _rel(y, yr) < 2e-5
    for got, want, name in zip(dev, ref_in, ('dx', 'dw1', 'db1', 'dw2', 'db2')):
        assert _rel(got.grad, want.grad) < 1e-4, name


def test_composed_resblock_equals_fused():
    """The general-conv composition (TTRAP_FUSED=0 path) and the fused kernels agree."""
    from timbre_trap.framework import ops
    C, d = 8, 2
    x, w1, b1 = _rand(2, C, 11, 50, seed=1), _rand(C, C, 3, 3, seed=2, scale=0.2), _rand(C, seed=3, scale=0.2)
    w2, b2 = _rand(C, C, 1, 1, seed=4, scale=0.3), _rand(C, seed=5, scale=0.2)
    a = [t.cuda().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    b = [t.cuda().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    ya = ops.residual_block(*a, d)
    ops.FUSED_RESBLOCK = False
    try:
        yb = ops.residual_block(*b, d)
    finally:
        ops.FUSED_RESBLOCK = True
    gy = _rand(2, C, 11, 50, seed=9).cuda()
    ya.backward(gy)
    yb.backward(gy)
    assert _rel(ya, yb) < 1e-5
    for p, q in zip(a, b):
        assert _rel(p.grad, q.grad) < 1e-4


CASES = [
    # name, kind, Cin, Cout, KH, KW, stride, dil, pad, out_pad, act, H, T
    ('convin', 'conv', 2, 4, 3, 3, 1, 1, 1, 0, 1, 12, 70),
    ('convout', 'conv', 4, 2, 3, 3, 1, 1, 1, 0, 0, 9, 33),
    ('res3x3_d3', 'conv', 8, 8, 3, 3, 1, 3, 3, 0, 1, 10, 40),
    ('res1x1', 'conv', 16, 16, 1, 1, 1, 1, 0, 0, 1, 7, 65),
    ('sconv', 'conv', 4, 8, 4, 1, 2, 1, 0, 0, 1, 13, 66),
    ('sconv_even', 'conv', 32, 64, 4, 1, 2, 1, 0, 0, 1, 64, 20),
    ('tconv_p0', 'tconv', 8, 4, 4, 1, 2, 1, 0, 0, 1, 6, 40),
    ('tconv_p1', 'tconv', 64, 32, 4, 1, 2, 1, 0, 1, 1, 5, 70),
    ('c2_mc1', 'conv', 2, 2, 3, 3, 1, 2, 2, 0, 1, 8, 30),
]


@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_general_conv_forward_backward(case):
    from timbre_trap.framework import ops
    _, kind, Cin, Cout, KH, KW, stride, dil, pad, out_pad, act, H, T = case
    x = _rand(2, Cin, H, T, seed=1)
    if kind == 'conv':
        w = _rand(Cout, Cin, KH, KW, seed=2, scale=0.3)
    else:
        w = _rand(Cin, Cout, KH, KW, seed=2, scale=0.3)
    b = _rand(Cout, seed=3, scale=0.2)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    if kind == 'conv':
        yr = F.conv2d(xr, wr, br, stride=(stride, 1), padding=(pad, pad if KW > 1 else 0), dilation=(dil, dil if KW > 1 else 1))
    else:
        yr = F.conv_transpose2d(xr, wr, br, stride=(stride, 1), output_padding=(out_pad, 0))
    if act:
        yr = F.elu(yr)
    gy = _rand(*yr.shape, seed=4)
    yr.backward(gy.double())
    xd, wd, bd = (t.cuda().requires_grad_(True) for t in (x, w, b))
    cfg = ops.ConvCfg(KH, KW, stride, dil, pad, pad if KW > 1 else 0, kind, out_pad, act)
    y = ops.conv(xd, wd, bd, cfg)
    assert y.shape == yr.shape
    y.backward(gy.cuda())
    assert _rel(y, yr) < 2e-5
    assert _rel(xd.grad, xr.grad) < 1e-4
    assert _rel(wd.grad, wr.grad) < 1e-4
    assert _rel(bd.grad, br.grad) < 1e-4


@pytest.mark.parametrize('C,D,E,T', [(64, 128, 31, 70), (32, 32, 31, 5), (8, 16, 3, 130)])
def test_latent_layers(C, D, E, T):
    from timbre_trap.framework import ops
    x = _rand(2, C, E, T, seed=1)
    w = _rand(D, C, E, 1, seed=2, scale=0.05)
    b = _rand(D, seed=3, scale=0.2)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = F.conv2d(xr, wr, br).squeeze(-2)
    gy = _rand(*yr.shape, seed=4)
    yr.backward(gy.double())
    xd, wd, bd = (t.cuda().requires_grad_(True) for t in (x, w, b))
    y = ops.LatentEncodeFn.apply(xd, wd, bd)
    y.backward(gy.cuda())
    assert _rel(y, yr) < 2e-5 and _rel(xd.grad, xr.grad) < 1e-4 and _rel(wd.grad, wr.grad) < 1e-4 and _rel(bd.grad, br.grad) < 1e-4

    z = _rand(2, D + 1, T, seed=5)
    w = _rand(D + 1, C, E, 1, seed=6, scale=0.1)
    b = _rand(C, seed=7, scale=0.2)
    zr, wr, br = (t.double().requires_grad_(True) for t in (z, w, b))
    yr = F.elu(F.conv_transpose2d(zr.unsqueeze(-2), wr, br))
    gy = _rand(*yr.shape, seed=8)
    yr.backward(gy.double())
    zd, wd, bd = (t.cuda().requires_grad_(True) for t in (z, w, b))
    y = ops.LatentDecodeFn.apply(zd, wd, bd)
    y.backward(gy.cuda())
    assert y.shape == yr.shape
    assert _rel(y, yr) < 2e-5 and _rel(zd.grad, zr.grad) < 1e-4 and _rel(wd.grad, wr.grad) < 1e-4 and _rel(bd.grad, br.grad) < 1e-4


def test_blocks_against_reference_golden(golden):
    """HIP modules loaded with the fixture weights reproduce the REFERENCE's recorded outputs."""
    from timbre_trap.framework import DecoderBlock, EncoderBlock, ResidualConv2dBlock
    g = golden('blocks')

    def load(mod, prefix):
        sd = {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}
        mod.load_state_dict(sd, strict=True)
        return mod.cuda()
    x = torch.from_numpy(g['res_x']).cuda()
    for d in (1, 2, 3):
        m = load(ResidualConv2dBlock(4, 4, 3, d), f'res_d{d}_sd.')
        np.testing.assert_allclose(m(x).detach().cpu().numpy(), g[f'res_d{d}_y'], rtol=2e-5, atol=2e-5)
    m = load(EncoderBlock(2, 4), 'encblk_sd.')
    np.testing.assert_allclose(m(torch.from_numpy(g['encblk_x']).cuda()).detach().cpu().numpy(), g['encblk_y'], rtol=2e-5, atol=2e-5)
    for p in (0, 1):
        m = load(DecoderBlock(4, 2, padding=p), f'decblk_p{p}_sd.')
        np.testing.assert_allclose(m(torch.from_numpy(g['decblk_x']).cuda()).detach().cpu().numpy(), g[f'decblk_p{p}_y'], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('C', [4, 8, 16, 32])
@pytest.mark.parametrize('H,T', [(13, 70), (20, 64), (9, 130), (22, 192), (37, 132)])
def test_mfma_strided_and_transposed_conv(C, H, T):
    """EncoderBlock.sconv / DecoderBlock.tconv on the MFMA kernels (Down4 / Up4 policies) vs float64 torch."""
    from timbre_trap.framework import ops
    x = _rand(2, C, H, T, seed=1)
    w = _rand(2 * C, C, 4, 1, seed=2, scale=0.5 / C ** 0.5)
    b = _rand(2 * C, seed=3, scale=0.3)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = F.elu(F.conv2d(xr, wr, br, stride=(2, 1)))
    gy = _rand(*yr.shape, seed=4)
    yr.backward(gy.double())
    xd, wd, bd = (t.cuda().requires_grad_(True) for t in (x, w, b))
    y = ops.strided_conv(xd, wd, bd, 4, 2)
    assert y.shape == yr.shape and isinstance(y.grad_fn, ops.StridedConvFn._backward_cls)
    y.backward(gy.cuda())
    assert _rel(y, yr) < 2e-5
    assert _rel(xd.grad, xr.grad) < 1e-4 and _rel(wd.grad, wr.grad) < 1e-4 and _rel(bd.grad, br.grad) < 1e-4

    for out_pad in (0, 1):
        x2 = _rand(2, 2 * C, H, T, seed=5)
        w2 = _rand(2 * C, C, 4, 1, seed=6, scale=0.5 / C ** 0.5)
        b2 = _rand(C, seed=7, scale=0.3)
        xr, wr, br = (t.double().requires_grad_(True) for t in (x2, w2, b2))
        yr = F.elu(F.conv_transpose2d(xr, wr, br, stride=(2, 1), output_padding=(out_pad, 0)))
        gy = _rand(*yr.shape, seed=8)
        yr.backward(gy.double())
        xd, wd, bd = (t.cuda().requires_grad_(True) for t in (x2, w2, b2))
        y = ops.transposed_conv(xd, wd, bd, 4, 2, out_pad)
        assert y.shape == yr.shape and isinstance(y.grad_fn, ops.TransposedConvFn._backward_cls)
        y.backward(gy.cuda())
        assert _rel(y, yr) < 2e-5
        assert _rel(xd.grad, xr.grad) < 1e-4 and _rel(wd.grad, wr.grad) < 1e-4 and _rel(bd.grad, br.grad) < 1e-4


@pytest.mark.parametrize('C,d', [(16, 1), (16, 3), (32, 1), (32, 2), (32, 3)])
@pytest.mark.parametrize('shape', [(2, 13, 128), (1, 9, 64)])
def test_resblock_bf16_operand_mode(C, d, shape, monkeypatch):
    """
    Opt-in precision mode: 3x3 conv operands (forward, data gradient) and dW1 rounded to bf16 on the matrix cores,
    fp32 accumulation, fp32 tensors.  Tolerance = bf16 rounding (2^-8 per operand) over K = 9C terms.
    """
    from timbre_trap.framework import ops
    monkeypatch.setattr(ops, 'PRECISION', 'bf16')
    B, H, T = shape
    x = _rand(B, C, H, T, seed=1)
    w1 = _rand(C, C, 3, 3, seed=2, scale=1.0 / (3 * C ** 0.5))
    b1 = _rand(C, seed=3, scale=0.3)
    w2 = _rand(C, C, 1, 1, seed=4, scale=1.0 / C ** 0.5)
    b2 = _rand(C, seed=5, scale=0.3)
    gy = _rand(B, C, H, T, seed=6)
    ref_in = [t.double().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    sd = {'p.conv1.0.weight': ref_in[1], 'p.conv1.0.bias': ref_in[2], 'p.conv2.0.weight': ref_in[3], 'p.conv2.0.bias': ref_in[4]}
    yr = oae.residual_block(ref_in[0], sd, 'p', d)
    yr.backward(gy.double())
    dev = [t.cuda().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    y = ops.residual_block(*dev, d)
    y.backward(gy.cuda())
    e = _rel(y, yr)
    assert 1e-5 < e < 1e-2, e                       # really bf16 (not the fp32 path), and within bf16 rounding
    for got, want, name in zip(dev, ref_in, ('dx', 'dw1', 'db1', 'dw2', 'db2')):
        assert _rel(got.grad, want.grad) < 2e-2, name


@pytest.mark.parametrize('C,d', [(16, 1), (16, 3), (32, 1), (32, 2), (32, 3)])
@pytest.mark.parametrize('shape', [(2, 13, 128), (1, 9, 64)])
def test_resblock_split_bf16_mode(C, d, shape, monkeypatch):
    """
    Split-bf16 ("bf16x3") mode: operands fed as hi + lo bf16 pairs, three matrix instructions per product block, fp32
    accumulation.  Stays inside the 1e-4 parity bar of the fp32 path (measured ~1e-6 against the float64 oracle).
    """
    from timbre_trap.framework import ops
    monkeypatch.setattr(ops, 'PRECISION', 'bf16x3')
    B, H, T = shape
    x = _rand(B, C, H, T, seed=1)
    w1 = _rand(C, C, 3, 3, seed=2, scale=1.0 / (3 * C ** 0.5))
    b1 = _rand(C, seed=3, scale=0.3)
    w2 = _rand(C, C, 1, 1, seed=4, scale=1.0 / C ** 0.5)
    b2 = _rand(C, seed=5, scale=0.3)
    gy = _rand(B, C, H, T, seed=6)
    ref_in = [t.double().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    sd = {'p.conv1.0.weight': ref_in[1], 'p.conv1.0.bias': ref_in[2], 'p.conv2.0.weight': ref_in[3], 'p.conv2.0.bias': ref_in[4]}
    yr = oae.residual_block(ref_in[0], sd, 'p', d)
    yr.backward(gy.double())
    dev = [t.cuda().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    y = ops.residual_block(*dev, d)
    y.backward(gy.cuda())
    e = _rel(y, yr)
    print('split-bf16 forward rel err', e)
    assert e < 2e-5, e
    for got, want, name in zip(dev, ref_in, ('dx', 'dw1', 'db1', 'dw2', 'db2')):
        assert _rel(got.grad, want.grad) < 2e-5, (name, _rel(got.grad, want.grad))


def test_bias_gradient_with_frozen_weight():
    """A trainable bias under a frozen weight still receives its gradient (weight / bias gradients gated separately)."""
    from timbre_trap.framework import ops
    for kind, wshape in (('conv', (4, 2, 3, 3)), ('tconv', (8, 4, 4, 1))):
        Cin = wshape[1] if kind == 'conv' else wshape[0]
        Cout = wshape[0] if kind == 'conv' else wshape[1]
        x = _rand(2, Cin, 12, 40, seed=1)
        w, b = _rand(*wshape, seed=2, scale=0.3), _rand(Cout, seed=3, scale=0.2)
        xr, wr, br = x.double(), w.double(), b.double().requires_grad_(True)
        if kind == 'conv':
            yr = F.elu(F.conv2d(xr, wr, br, padding=1))
            cfg = ops.ConvCfg(3, 3, 1, 1, 1, 1, 'conv', 0, 1)
        else:
            yr = F.elu(F.conv_transpose2d(xr, wr, br, stride=(2, 1)))
            cfg = ops.ConvCfg(4, 1, 2, 1, 0, 0, 'tconv', 0, 1)
        gy = _rand(*yr.shape, seed=4)
        yr.backward(gy.double())
        xd, wd, bd = x.cuda(), w.cuda(), b.cuda().requires_grad_(True)
        y = ops.conv(xd, wd, bd, cfg)
        y.backward(gy.cuda())
        assert wd.grad is None and _rel(bd.grad, br.grad) < 1e-4


def test_four_pixel_narrow_forward_is_bit_identical(tmp_path):
    """
    k_small_fwd4 (csrc/conv_small.hip: four frames per lane, 16-byte tap reads and stores) performs the same products in the same order
    on the same instruction as k_small_lds: y and the saved hidden activation must be BIT-identical for C = 4, 8 x three dilations x
    ragged and bench-sized planes.  The switch is read once per process, so each setting runs in its own child
    (TTRAP_SMALL_FWD4 = 0: never, 2: every shape; the default takes it where it wins).
    """
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, 'tools', 'diag', 'small_fwd_check.py')
    outs = []
    for setting in ('0', '2'):
        out = str(tmp_path / ('fwd4_%s.pt' % setting))
        env = dict(os.environ, TTRAP_SMALL_FWD4=setting)
        r = subprocess.run([sys.executable, tool, out], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(out)
    r = subprocess.run([sys.executable, tool, 'cmp'] + outs, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and '0 differ' in r.stdout, r.stdout[-2000:]
