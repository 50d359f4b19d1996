"""
GPU parity in the launch regime bench.py runs the kernels in: persistent workgroups that each walk SEVERAL tiles
(cross-tile LDS-DMA prefetch into the other buffer, XCD-ordered tile walk with a tail when ntiles % 8 != 0, partial
weight-gradient images summed by the second-stage reduce).  Two routes:

  * bench launch shapes (> 512 tiles: C = 32 at B 8 x H 65 x T 1024, C = 16 at B 5 x H 133 x T 1088, the narrow levels at
    their real heights) against the float64 oracle at the tolerances of tests/test_gpu_conv.py;
  * small shapes with the persistent grids capped through tt_set_cu_limit (include/ttrap.h) so that every workgroup walks
    many tiles -- cheap enough to sweep dilations, both backward modes, ragged edges and ntiles % 8 tails.
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


@pytest.fixture
def cu_limit():
    """Cap the CU count the persistent grids are sized with; restored afterwards."""
    from timbre_trap import _hip
    lib = _hip.lib()
    prev = lib.tt_set_cu_limit(0)

    def set_(n):
        lib.tt_set_cu_limit(n)
    yield set_
    lib.tt_set_cu_limit(prev)


def _resblock_case(C, d, B, H, T, save_hidden, monkeypatch, fwd_tol=2e-5, grad_tol=1e-4):
    from timbre_trap.framework import ops
    monkeypatch.setattr(ops, 'SAVE_HIDDEN', save_hidden)
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
    assert _rel(y, yr) < fwd_tol
    for got, want, name in zip(dev, ref_in, ('dx', 'dw1', 'db1', 'dw2', 'db2')):
        assert _rel(got.grad, want.grad) < grad_tol, name


# ---- bench launch shapes ------------------------------------------------------------------------------------------------

BENCH_SHAPES = [
    # C, d, B, H, T            tiles (8 x 64 for C >= 16; 16 x 64 for C <= 8)
    (32, 1, 8, 65, 1024),     # 1152 tiles on 512 workgroups: the bench kernel k_rb_fwd<32,1>, 2-3 tiles per workgroup
    (32, 3, 8, 65, 1024),
    (16, 2, 5, 133, 1088),    # 5 * 17 * 17 = 1445 tiles, 1445 % 8 = 5: xcd_tile tail
    (8, 3, 5, 269, 1024),     # 5 * 17 * 16 = 1360 tiles of the narrow LDS kernel on 256 workgroups
    (4, 2, 5, 540, 1024),     # 5 * 34 * 16 = 2720 tiles on 512 workgroups
]


@pytest.mark.parametrize('C,d,B,H,T', BENCH_SHAPES)
@pytest.mark.parametrize('save_hidden', [True, False])
def test_resblock_bench_launch_shapes(C, d, B, H, T, save_hidden, monkeypatch):
    _resblock_case(C, d, B, H, T, save_hidden, monkeypatch)


@pytest.mark.parametrize('C,B,H,T', [(32, 8, 65, 1024), (16, 5, 133, 1088), (8, 5, 269, 1024), (4, 3, 540, 1024)])
def test_strided_transposed_bench_launch_shapes(C, B, H, T):
    """EncoderBlock.sconv at its real input height, DecoderBlock.tconv back up (Down4 / Up4 and each other's gradients)."""
    from timbre_trap.framework import ops
    x = _rand(B, C, H, T, seed=1)
    w = _rand(2 * C, C, 4, 1, seed=2, scale=0.5 / C ** 0.5)
    b = _rand(2 * C, seed=3, scale=0.3)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = F.elu(F.conv2d(xr, wr, br, stride=(2, 1)))
    gy = _rand(*yr.shape, seed=4)
    yr.backward(gy.double())
    xd, wd, bd = (t.cuda().requires_grad_(True) for t in (x, w, b))
    y = ops.strided_conv(xd, wd, bd, 4, 2)
    y.backward(gy.cuda())
    assert _rel(y, yr) < 2e-5
    assert _rel(xd.grad, xr.grad) < 1e-4 and _rel(wd.grad, wr.grad) < 1e-4 and _rel(bd.grad, br.grad) < 1e-4

    Hs = yr.shape[2]
    out_pad = H % 2
    x2 = _rand(B, 2 * C, Hs, T, seed=5)
    w2 = _rand(2 * C, C, 4, 1, seed=6, scale=0.5 / C ** 0.5)
    b2 = _rand(C, seed=7, scale=0.3)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x2, w2, b2))
    yr = F.elu(F.conv_transpose2d(xr, wr, br, stride=(2, 1), output_padding=(out_pad, 0)))
    assert yr.shape[2] == H
    gy = _rand(*yr.shape, seed=8)
    yr.backward(gy.double())
    xd, wd, bd = (t.cuda().requires_grad_(True) for t in (x2, w2, b2))
    y = ops.transposed_conv(xd, wd, bd, 4, 2, out_pad)
    y.backward(gy.cuda())
    assert _rel(y, yr) < 2e-5
    assert _rel(xd.grad, xr.grad) < 1e-4 and _rel(wd.grad, wr.grad) < 1e-4 and _rel(bd.grad, br.grad) < 1e-4


@pytest.mark.parametrize('B', [8, 19])
def test_latent_heads_bench_batch(B):
    """Latent GEMMs at B >= 8 (batch-reduced weight-gradient form, several clips per workgroup column)."""
    from timbre_trap.framework import ops
    C, D, E, T = 64, 128, 31, 1024 if B == 8 else 132
    x = _rand(B, C, E, T, seed=1)
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
    z = _rand(B, D + 1, T, seed=5)
    w = _rand(D + 1, C, E, 1, seed=6, scale=0.1)
    b = _rand(C, seed=7, scale=0.2)
    zr, wr, br = (t.double().requires_grad_(True) for t in (z, w, b))
    yr = F.elu(F.conv_transpose2d(zr.unsqueeze(-2), wr, br))
    gy = _rand(*yr.shape, seed=8)
    yr.backward(gy.double())
    zd, wd, bd = (t.cuda().requires_grad_(True) for t in (z, w, b))
    y = ops.LatentDecodeFn.apply(zd, wd, bd)
    y.backward(gy.cuda())
    assert _rel(y, yr) < 2e-5 and _rel(zd.grad, zr.grad) < 1e-4 and _rel(wd.grad, wr.grad) < 1e-4 and _rel(bd.grad, br.grad) < 1e-4


# ---- capped grids: many tiles per workgroup on small shapes ----------------------------------------------------------------

@pytest.mark.parametrize('C,d', [(4, 1), (4, 3), (8, 1), (8, 2), (16, 1), (16, 2), (16, 3), (32, 1), (32, 2), (32, 3)])
@pytest.mark.parametrize('shape', [(3, 21, 200), (2, 70, 132), (5, 9, 68), (1, 37, 324)])
@pytest.mark.parametrize('save_hidden', [True, False])
@pytest.mark.parametrize('cus', [1, 3])
def test_resblock_capped_grid(C, d, shape, save_hidden, cus, monkeypatch, cu_limit):
    cu_limit(cus)
    B, H, T = shape
    _resblock_case(C, d, B, H, T, save_hidden, monkeypatch)


@pytest.mark.parametrize('C', [4, 8, 16, 32])
@pytest.mark.parametrize('H,T', [(37, 132), (70, 200), (22, 324)])
@pytest.mark.parametrize('cus', [1, 3])
def test_strided_transposed_capped_grid(C, H, T, cus, cu_limit):
    from timbre_trap.framework import ops
    cu_limit(cus)
    x = _rand(3, C, H, T, seed=1)
    w = _rand(2 * C, C, 4, 1, seed=2, scale=0.5 / C ** 0.5)
    b = _rand(2 * C, seed=3, scale=0.3)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = F.elu(F.conv2d(xr, wr, br, stride=(2, 1)))
    gy = _rand(*yr.shape, seed=4)
    yr.backward(gy.double())
    xd, wd, bd = (t.cuda().requires_grad_(True) for t in (x, w, b))
    y = ops.strided_conv(xd, wd, bd, 4, 2)
    y.backward(gy.cuda())
    assert _rel(y, yr) < 2e-5
    assert _rel(xd.grad, xr.grad) < 1e-4 and _rel(wd.grad, wr.grad) < 1e-4 and _rel(bd.grad, br.grad) < 1e-4
    for out_pad in (0, 1):
        x2 = _rand(3, 2 * C, H, T, seed=5)
        w2 = _rand(2 * C, C, 4, 1, seed=6, scale=0.5 / C ** 0.5)
        b2 = _rand(C, seed=7, scale=0.3)
        xr, wr, br = (t.double().requires_grad_(True) for t in (x2, w2, b2))
        yr = F.elu(F.conv_transpose2d(xr, wr, br, stride=(2, 1), output_padding=(out_pad, 0)))
        gy = _rand(*yr.shape, seed=8)
        yr.backward(gy.double())
        xd, wd, bd = (t.cuda().requires_grad_(True) for t in (x2, w2, b2))
        y = ops.transposed_conv(xd, wd, bd, 4, 2, out_pad)
        y.backward(gy.cuda())
        assert _rel(y, yr) < 2e-5
        assert _rel(xd.grad, xr.grad) < 1e-4 and _rel(wd.grad, wr.grad) < 1e-4 and _rel(bd.grad, br.grad) < 1e-4


@pytest.mark.parametrize('cus', [1, 3])
def test_boundary_convs_capped_grid(cus, cu_limit):
    """The 3x3 in / out convolutions (2 -> 4, 4 -> 2) of the encoder / decoder with several tiles per workgroup."""
    from timbre_trap.framework import ops
    cu_limit(cus)
    for Cin, Cout, act in ((2, 4, 1), (4, 2, 0)):
        x = _rand(3, Cin, 45, 260, seed=1)
        w = _rand(Cout, Cin, 3, 3, seed=2, scale=0.3)
        b = _rand(Cout, seed=3, scale=0.2)
        xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
        yr = F.conv2d(xr, wr, br, padding=1)
        if act:
            yr = F.elu(yr)
        gy = _rand(*yr.shape, seed=4)
        yr.backward(gy.double())
        xd, wd, bd = (t.cuda().requires_grad_(True) for t in (x, w, b))
        y = ops.conv(xd, wd, bd, ops.ConvCfg(3, 3, 1, 1, 1, 1, 'conv', 0, act))
        y.backward(gy.cuda())
        assert _rel(y, yr) < 2e-5
        assert _rel(xd.grad, xr.grad) < 1e-4 and _rel(wd.grad, wr.grad) < 1e-4 and _rel(bd.grad, br.grad) < 1e-4


def test_capped_grid_equals_full_grid_bitwise_forward(cu_limit):
    """The tile walk must not change values: a forward pass is bit-identical whatever the number of workgroups."""
    from timbre_trap.framework import ops
    for C in (4, 8, 16, 32):
        x = _rand(3, C, 29, 200, seed=1).cuda()
        w1, b1 = _rand(C, C, 3, 3, seed=2, scale=0.2).cuda(), _rand(C, seed=3, scale=0.2).cuda()
        w2, b2 = _rand(C, C, 1, 1, seed=4, scale=0.3).cuda(), _rand(C, seed=5, scale=0.2).cuda()
        cu_limit(256)
        full = ops.residual_block(x, w1, b1, w2, b2, 2)
        for cus in (1, 2, 5):
            cu_limit(cus)
            assert torch.equal(ops.residual_block(x, w1, b1, w2, b2, 2), full), (C, cus)


@pytest.mark.parametrize('C', [4, 8])
@pytest.mark.parametrize('d', [1, 2, 3])
@pytest.mark.parametrize('cus', [2, 256])
def test_fused_narrow_backward_matches_three_kernel_path(C, d, cus, monkeypatch, cu_limit):
    """
    The one-pass narrow-level backward (pointwise chain + data gradient + packed MFMA weight gradient, csrc/conv_small.hip
    k_small_bwd_fused; default at C = 4, opt-in at C = 8) against the three-kernel path and the float64 oracle, with several
    tiles per workgroup (ragged edges, ntiles % 8 != 0).
    """
    from timbre_trap.framework import ops
    monkeypatch.setattr(ops, 'SAVE_HIDDEN', True)
    cu_limit(cus)
    B, H, T = 3, 45, 200
    x = _rand(B, C, H, T, seed=1)
    w1 = _rand(C, C, 3, 3, seed=2, scale=1.0 / (3 * C ** 0.5))
    b1 = _rand(C, seed=3, scale=0.3)
    w2 = _rand(C, C, 1, 1, seed=4, scale=1.0 / C ** 0.5)
    b2 = _rand(C, seed=5, scale=0.3)
    gy = _rand(B, C, H, T, seed=6)
    ref_in = [t.double().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    sd = {'p.conv1.0.weight': ref_in[1], 'p.conv1.0.bias': ref_in[2], 'p.conv2.0.weight': ref_in[3], 'p.conv2.0.bias': ref_in[4]}
    oae.residual_block(ref_in[0], sd, 'p', d).backward(gy.double())
    grads = {}
    for mode in ('fused', 'three_kernel'):
        monkeypatch.delenv('TTRAP_SMALL_UNFUSED_BWD', raising=False)
        monkeypatch.delenv('TTRAP_SMALL_FUSED_BWD_C8', raising=False)
        if mode == 'fused':
            monkeypatch.setenv('TTRAP_SMALL_FUSED_BWD_C8', '1')
        else:
            monkeypatch.setenv('TTRAP_SMALL_UNFUSED_BWD', '1')
        dev = [t.cuda().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
        ops.residual_block(*dev, d).backward(gy.cuda())
        grads[mode] = [t.grad.clone() for t in dev]
        for got, want, name in zip(dev, ref_in, ('dx', 'dw1', 'db1', 'dw2', 'db2')):
            assert _rel(got.grad, want.grad) < 1e-4, (mode, name)
    for a, b, name in zip(grads['fused'], grads['three_kernel'], ('dx', 'dw1', 'db1', 'dw2', 'db2')):
        assert _rel(a, b) < 2e-5, name
    # the two paths really are different code (the summation orders differ): at least one gradient is not bit-identical
    assert any(not torch.equal(a, b) for a, b in zip(grads['fused'], grads['three_kernel']))


@pytest.mark.parametrize('C', [4, 8])
@pytest.mark.parametrize('d', [1, 2, 3])
def test_narrow_matrix_pipe_form_is_bit_identical_to_the_vector_form(C, d, monkeypatch):
    """
    k_small_lds issues the narrow 3x3 convolution either as per-lane FMAs or as v_mfma_f32_4x4x1 (lane = pixel, accumulator
    register = output channel): one FMA per product in the same order, so forward and data gradient must agree BITWISE.
    """
    from timbre_trap.framework import ops
    monkeypatch.setattr(ops, 'SAVE_HIDDEN', True)
    monkeypatch.setenv('TTRAP_SMALL_UNFUSED_BWD', '1')              # data gradient through k_small_lds as well
    x = _rand(3, C, 45, 200, seed=1).cuda()
    w1, b1 = _rand(C, C, 3, 3, seed=2, scale=0.2).cuda(), _rand(C, seed=3, scale=0.2).cuda()
    w2, b2 = _rand(C, C, 1, 1, seed=4, scale=0.3).cuda(), _rand(C, seed=5, scale=0.2).cuda()
    outs = {}
    for form in ('matrix', 'vector'):
        if form == 'vector':
            monkeypatch.setenv('TTRAP_SMALL_VALU_FMA', '1')
        else:
            monkeypatch.delenv('TTRAP_SMALL_VALU_FMA', raising=False)
        xr = x.clone().requires_grad_(True)
        y = ops.ResBlockFn.apply(xr, w1, b1, w2, b2, d)
        gx, = torch.autograd.grad(y, xr, torch.ones_like(y))
        outs[form] = (y.detach().clone(), gx.clone())
    assert torch.equal(outs['matrix'][0], outs['vector'][0]) and torch.equal(outs['matrix'][1], outs['vector'][1])
