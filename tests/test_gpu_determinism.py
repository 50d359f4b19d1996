"""
Run-to-run reproducibility of the backward entry points at the bench heights: the same call on the same buffers must give the same
bits (outputs whose reduce kernel adds with atomics -- the narrow levels' 3x3 weight gradient -- are compared at fp32 rounding instead).

Why this file exists: one build of the narrow one-pass backward returned ONE of its 64 pointwise-weight-gradient accumulators different
from run to run (a single pixel product wrong in ~3 % of the workgroups) while every tolerance test around it still passed most of the
time -- see the note at the accumulation loop of k_nrb_bwd_fused (csrc/conv_wide_bf16.hip) and tools/probes/dbg_nrb.py.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

HEIGHTS = {4: 540, 8: 269, 16: 133, 32: 65}


def _block_backward(C, d, runs=6, B=2, T=1024):
    from timbre_trap import _hip
    from timbre_trap._hip import check, ptr, stream_ptr
    lib, st = _hip.lib(), stream_ptr()
    H = HEIGHTS[C]
    g = torch.Generator(device='cuda').manual_seed(7)
    rnd = lambda *s, scale=1.0: torch.randn(*s, device='cuda', generator=g) * scale
    w1, b1, w2, b2 = rnd(C, C, 3, 3, scale=0.1), rnd(C, scale=0.1), rnd(C, C, 1, 1, scale=0.3), rnd(C, scale=0.1)
    xb, gb = rnd(B, H, T, C).bfloat16(), rnd(B, H, T, C).bfloat16()
    yb, hb, dxb = (torch.empty_like(xb) for _ in range(3))
    check(lib.tt_wide_rb_fwd(ptr(xb), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(yb), ptr(hb), B, C, H, T, d, st), 'fwd')
    ws = torch.zeros(lib.tt_wide_scratch_bytes(B, C, H, T), dtype=torch.uint8, device='cuda')
    out = []
    for _ in range(runs):
        grads = [torch.zeros(s, dtype=torch.float32, device='cuda') for s in ((C, C, 3, 3), (C,), (C, C, 1, 1), (C,))]
        check(lib.tt_wide_rb_bwd(ptr(xb), ptr(hb), ptr(gb), ptr(w1), ptr(w2), ptr(b2), ptr(dxb), ptr(grads[0]), ptr(grads[1]),
                                 ptr(grads[2]), ptr(grads[3]), ptr(ws), B, C, H, T, d, st), 'bwd')
        torch.cuda.synchronize()
        out.append([t.clone() for t in grads] + [dxb.clone()])
    return out


@pytest.mark.parametrize('C,d', [(8, 1), (8, 2), (8, 3), (4, 1), (4, 2), (4, 3), (16, 1), (16, 3), (32, 2)])
def test_block_backward_is_reproducible(C, d):
    runs = _block_backward(C, d)
    for r in runs[1:]:
        for name, a, b in zip(('dw1', 'db1', 'dw2', 'db2', 'dx'), runs[0], r):
            if name == 'dw1' and C <= 8:            # k_nrb_reduce adds the 16 / C diagonal blocks of an element with atomicAdd
                assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()), name
            else:
                assert torch.equal(a, b), '%s differs between two runs of the same call (max %.3e)' % (name, float((a.float() - b.float()).abs().max()))


def test_train_step_gradients_are_reproducible():
    """The bench's own step (model_complexity 2 / latent 128, two clips x T = 1024, under autocast) computed twice from the same
    weights and audio: outputs and losses bit-identical, every parameter gradient bit-identical or equal at fp32 rounding."""
    from timbre_trap.framework import TimbreTrap, compute_consistency_loss, compute_reconstruction_loss, compute_transcription_loss
    torch.manual_seed(3)
    model = TimbreTrap(sample_rate=22050, n_octaves=9, bins_per_octave=60, secs_per_block=3, latent_size=128, model_complexity=2).cuda()
    g = torch.Generator().manual_seed(11)
    audio = (torch.rand(2, 1, 66150, generator=g) * 2 - 1).cuda()
    target = (torch.rand(2, 540, 1024, generator=g) > 0.97).float().cuda()

    def once():
        coefficients = model.sliCQ(audio)
        with torch.autocast(device_type='cuda', dtype=torch.bfloat16):
            reconstruction, latents, trn_coeffs, trn_rec, trn_scr, _ = model(audio, True)
            transcription = model.to_activations(trn_coeffs)
            l_rec = compute_reconstruction_loss(reconstruction, coefficients)
            l_trn = compute_transcription_loss(transcription, target, True)
            l_sp, l_sc = compute_consistency_loss(trn_rec, trn_scr, trn_coeffs)
            total = l_rec + l_trn + (l_sp + l_sc)
            model.zero_grad()
            total.backward()
        torch.cuda.synchronize()
        outs = dict(reconstruction=reconstruction, latents=latents, trn=trn_coeffs, trn_rec=trn_rec, trn_scr=trn_scr,
                    losses=torch.stack([l_rec, l_trn, l_sp, l_sc]).float())
        return {k: v.detach().float().clone() for k, v in outs.items()}, {k: p.grad.detach().clone() for k, p in model.named_parameters()}

    o1, g1 = once()
    o2, g2 = once()
    for k in o1:
        assert torch.equal(o1[k], o2[k]), 'output %s differs between two identical steps' % k
    # gradients: bit-identical, or at fp32 rounding where a reduce adds with atomics (the narrow blocks' 3x3 weight gradients in
    # k_nrb_reduce, the latent heads' bias sums) -- a wrong pixel product, as in the failure this file was written after, is 1e-3
    for k in g1:
        if not torch.equal(g1[k], g2[k]):
            rel = float((g1[k] - g2[k]).abs().max() / (g1[k].abs().max() + 1e-30))
            assert rel < 2e-6, (k, tuple(g1[k].shape), rel)


# ---- every other backward C entry point of the bf16 path at the bench heights (round-3 verdict, item 7) --------------------------------
# Bitwise wherever the partial sums are reduced in a fixed order (register dumps + k_*_reduce); fp32 rounding where the kernel ends in
# atomicAdd (the latent heads' bias gradient, latent_bf16.hip).  A wrong pixel product -- the failure this file was written after --
# shows at 1e-3 of the gradient and fails either comparison.

def _same(runs, names, atomic=()):
    for r in runs[1:]:
        for name, a, b in zip(names, runs[0], r):
            if name in atomic:
                assert float((a.float() - b.float()).abs().max()) <= 2e-6 * float(a.float().abs().max()) + 1e-30, name
            else:
                assert torch.equal(a, b), '%s differs between two runs of the same call (max %.3e of %.3e)' % (
                    name, float((a.float() - b.float()).abs().max()), float(a.float().abs().max()))


@pytest.mark.parametrize('C', [4, 8, 16, 32])
@pytest.mark.parametrize('kind', ['sconv', 'tconv', 'sconv-pregated', 'tconv-pregated'])
def test_strided_backward_is_reproducible(C, kind, runs=6, B=2, T=1024):
    from timbre_trap import _hip
    from timbre_trap._hip import check, ptr, stream_ptr
    from timbre_trap.framework import ops
    lib, st = _hip.lib(), stream_ptr()
    H = HEIGHTS[C]
    Ho = (H - 4) // 2 + 1
    pad = H - (2 * Ho + 2)
    big = ops.new_cl16(B, C, H, T, 'cuda').normal_()               # the C-channel side
    small = ops.new_cl16(B, 2 * C, Ho, T, 'cuda').normal_()        # the 2C-channel side
    w = torch.randn(2 * C, C, 4, 1, device='cuda') * 0.1
    ws = torch.empty(lib.tt_stride16_scratch_bytes(C), dtype=torch.uint8, device='cuda')
    out = []
    for _ in range(runs):
        if kind.startswith('sconv'):      # x = big, y / dy = small
            dx, dw, db = ops.new_cl16(B, C, H, T, 'cuda'), torch.zeros_like(w), torch.zeros(2 * C, device='cuda')
            y = small.clone()
            dy = ops.new_cl16(B, 2 * C, Ho, T, 'cuda').copy_(small.flip(0))
            if kind == 'sconv':
                check(lib.tt_sconv16_bwd(ptr(big), ptr(y), ptr(dy), ptr(w), ptr(dx), ptr(dw), ptr(db), ptr(ws), B, C, H, T, st), 'sconv bwd')
            else:                # round 5: dy taken as already gated (both operands by LDS-DMA, db as a matrix product)
                check(lib.tt_sconv16_bwd_pregated(ptr(big), ptr(dy), ptr(w), ptr(dx), ptr(dw), ptr(db), ptr(ws), B, C, H, T, st), 'sconv bwd pregated')
        else:                    # x = small, y / dy = big
            dx, dw, db = ops.new_cl16(B, 2 * C, Ho, T, 'cuda'), torch.zeros_like(w), torch.zeros(C, device='cuda')
            dy = ops.new_cl16(B, C, H, T, 'cuda').copy_(big.flip(0))
            if kind == 'tconv':
                check(lib.tt_tconv16_bwd(ptr(small), ptr(big), ptr(dy), ptr(w), ptr(dx), ptr(dw), ptr(db), ptr(ws), B, C, Ho, T, pad, st), 'tconv bwd')
            else:                # ... and, where the kernel has the form, with dx leaving gated for the latent head in front
                check(lib.tt_tconv16_bwd_pregated(ptr(small), ptr(dy), ptr(w), ptr(dx), ptr(dw), ptr(db), ptr(ws), B, C, Ho, T, pad,
                                                  1 if C in (16, 32) else 0, st), 'tconv bwd pregated')
        torch.cuda.synchronize()
        out.append([dx.clone(), dw.clone(), db.clone()])
    _same(out, ('dx', 'dw', 'db'))


@pytest.mark.parametrize('head', ['encoder.convlat', 'decoder.convin'])
def test_latent_heads_backward_is_reproducible(head, runs=6, B=2, T=1024):
    from timbre_trap.framework import ops
    CT, E, D = 64, 31, 128
    g = torch.Generator(device='cuda').manual_seed(5)
    rnd = lambda *s, scale=1.0: torch.randn(*s, device='cuda', generator=g) * scale
    top = ops.new_cl16(B, CT, E, T, 'cuda').copy_(rnd(B, CT, E, T))
    dtop = ops.new_cl16(B, CT, E, T, 'cuda').copy_(rnd(B, CT, E, T))
    out = []
    if head == 'encoder.convlat':
        w, b, dlat = rnd(D, CT, E, 1, scale=0.02), rnd(D, scale=0.1), rnd(B, D, T)
        for _ in range(runs):
            x16, wd, bd = top.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
            ops.LatEnc16Fn.apply(x16, wd, bd).backward(dlat)
            torch.cuda.synchronize()
            out.append([x16.grad.clone(), wd.grad.clone(), bd.grad.clone()])
        _same(out, ('dtop', 'dw', 'db'), atomic=('db',))
        out = []                                                 # round 5: the data gradient leaving gated (tt_latent16_expand_gated)
        for _ in range(runs):
            link = ops.GateLink()
            link.producer = True
            x16, wd, bd = top.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
            ops.LatEnc16Fn.apply(x16, wd, bd, link).backward(dlat)
            torch.cuda.synchronize()
            assert link.gated
            out.append([x16.grad.clone(), wd.grad.clone()])
        _same(out, ('dtop gated', 'dw'))
    else:
        w, b, z = rnd(D + 1, CT, E, 1, scale=0.02), rnd(CT, scale=0.1), rnd(B, D, T)
        for _ in range(runs):
            zd, wd, bd = z.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
            y = ops.LatDec16Fn.apply(zd, wd, bd, 1.0)
            y.backward(dtop)
            torch.cuda.synchronize()
            out.append([y.detach().clone(), zd.grad.clone(), wd.grad.clone(), bd.grad.clone()])
        _same(out, ('y', 'dz', 'dw', 'db'), atomic=('db',))
        out = []                                                 # round 5: from a gradient that arrives gated (tt_latent16_*_pregated)
        for _ in range(runs):
            link = ops.GateLink()
            zd, wd, bd = z.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
            y = ops.LatDec16Fn.apply(zd, wd, bd, 1.0, link)
            link.gated = True
            y.backward(dtop)
            torch.cuda.synchronize()
            out.append([zd.grad.clone(), wd.grad.clone(), bd.grad.clone()])
        _same(out, ('dz', 'dw', 'db'))           # round 6: the pregated bias gradient is summed over the rows in a fixed order (was E atomics per channel)


@pytest.mark.parametrize('which', ['convin', 'convout'])
def test_boundary_convs_backward_is_reproducible(which, runs=6, B=2, H=540, T=1024):
    from timbre_trap.framework import ops
    g = torch.Generator(device='cuda').manual_seed(6)
    rnd = lambda *s, scale=1.0: torch.randn(*s, device='cuda', generator=g) * scale
    out = []
    if which == 'convin':
        x, w, b = rnd(B, 2, H, T), rnd(4, 2, 3, 3, scale=0.3), rnd(4, scale=0.1)
        dy = ops.new_cl16(B, 4, H, T, 'cuda').copy_(rnd(B, 4, H, T))
        for _ in range(runs):
            xd, wd, bd = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
            ops.ConvIn16Fn.apply(xd, wd, bd).backward(dy)
            torch.cuda.synchronize()
            out.append([xd.grad.clone(), wd.grad.clone(), bd.grad.clone()])
    else:
        x4 = ops.new_cl16(B, 4, H, T, 'cuda').copy_(rnd(B, 4, H, T))
        w, b, dz = rnd(2, 4, 3, 3, scale=0.3), rnd(2, scale=0.1), rnd(B, 2, H, T)
        for _ in range(runs):
            xd, wd, bd = x4.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
            ops.ConvOut16Fn.apply(xd, wd, bd).backward(dz)
            torch.cuda.synchronize()
            out.append([xd.grad.clone(), wd.grad.clone(), bd.grad.clone()])
    _same(out, ('dx', 'dw', 'db'))


@pytest.mark.parametrize('C,d', [(16, 1), (16, 2), (16, 3), (32, 1), (32, 2), (32, 3)])
@pytest.mark.parametrize('entry', ['tt_wide_rb_bwd_onepass', 'tt_wide_rb_bwd_fused'])
def test_one_pass_wide_backward_is_reproducible(C, d, entry, runs=6, B=2, T=1024):
    """Both one-pass wide backward kernels (h1 saved: k_wrb_bwd1; h1 recomputed: k_wrb_bwd_fused) called directly."""
    from timbre_trap import _hip
    from timbre_trap._hip import check, ptr, stream_ptr
    lib, st = _hip.lib(), stream_ptr()
    H = HEIGHTS[C]
    g = torch.Generator(device='cuda').manual_seed(7)
    rnd = lambda *s, scale=1.0: torch.randn(*s, device='cuda', generator=g) * scale
    w1, b1, w2, b2 = rnd(C, C, 3, 3, scale=0.1), rnd(C, scale=0.1), rnd(C, C, 1, 1, scale=0.3), rnd(C, scale=0.1)
    xb, gb = rnd(B, H, T, C).bfloat16(), rnd(B, H, T, C).bfloat16()
    yb, hb, dxb = (torch.empty_like(xb) for _ in range(3))
    check(lib.tt_wide_rb_fwd(ptr(xb), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(yb), ptr(hb), B, C, H, T, d, st), 'fwd')
    ws = torch.zeros(lib.tt_wide_scratch_bytes(B, C, H, T), dtype=torch.uint8, device='cuda')
    out = []
    for _ in range(runs):
        gr = [torch.zeros(s, dtype=torch.float32, device='cuda') for s in ((C, C, 3, 3), (C,), (C, C, 1, 1), (C,))]
        if entry == 'tt_wide_rb_bwd_onepass':
            check(lib.tt_wide_rb_bwd_onepass(ptr(xb), ptr(hb), ptr(gb), ptr(w1), ptr(w2), ptr(b2), ptr(dxb), ptr(gr[0]), ptr(gr[1]), ptr(gr[2]),
                                             ptr(gr[3]), ptr(ws), B, C, H, T, d, st), entry)
        else:
            check(lib.tt_wide_rb_bwd_fused(ptr(xb), ptr(gb), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(dxb), ptr(gr[0]), ptr(gr[1]), ptr(gr[2]),
                                           ptr(gr[3]), ptr(ws), B, C, H, T, d, st), entry)
        torch.cuda.synchronize()
        out.append([t.clone() for t in gr] + [dxb.clone()])
    _same(out, ('dw1', 'db1', 'dw2', 'db2', 'dx'))


def test_split_operand_inference_chain_is_reproducible(runs=5, B=4, T=1024):
    """
    The split-operand forward kernels (csrc/conv_x3.hip) have no reduction across workgroups, but they accumulate with packed
    multiply-adds next to matrix instructions like the loop this file was written for -- so the same encoder-top / decoder-top chain
    (level 16 -> 16 -> 32 -> level 32 -> 32 -> 64 -> latent heads -> 64 -> 32 -> level 32 -> 32 -> 16 -> level 16) at the bench heights
    must return the same bits run after run.
    """
    import torch.nn as nn
    from timbre_trap.framework import modules, ops
    torch.manual_seed(3)
    enc3, enc4 = modules.EncoderBlock(16, 32).cuda(), modules.EncoderBlock(32, 64).cuda()
    dec1, dec2 = modules.DecoderBlock(64, 32, padding=1).cuda(), modules.DecoderBlock(32, 16, padding=1).cuda()
    convlat = nn.Conv2d(64, 128, (31, 1)).cuda()
    convin = nn.ConvTranspose2d(129, 64, (31, 1)).cuda()
    g = torch.Generator(device='cuda').manual_seed(5)
    x = torch.randn(B, 16, HEIGHTS[16], T, device='cuda', generator=g)
    outs = []
    with torch.no_grad(), ops.x3_chain_scope(True):
        for _ in range(runs):
            top = enc4(enc3(x, out_x3=True), out_x3=True)
            assert ops.is_x3(top) and top.shape == (B, 31, T, 2, 64)
            z = ops.latent_encode(top, convlat.weight, convlat.bias)
            y = ops.latent_decode(z, convin.weight, convin.bias, fill=1.0, out_x3=True)
            assert ops.is_x3(y)
            out = dec2(dec1(y, out_x3=True))
            assert out.dtype == torch.float32 and out.shape == x.shape
            outs.append((top.clone(), z.clone(), out.clone()))
    torch.cuda.synchronize()
    for r in outs[1:]:
        for a, b in zip(outs[0], r):
            assert torch.equal(a, b)
    assert bool(torch.isfinite(outs[0][2]).all())
