"""GPU parity of the HIP constant-Q transform (csrc/cqt.hip through tt_cqt_forward / tt_cqt_inverse) vs oracle/nsgt.py."""

import numpy as np
import pytest
import torch

from oracle import nsgt

pytestmark = pytest.mark.gpu
N, M, SR = 66150, 1024, 22050
REL = 1e-4          # north-star tolerance for CQT coefficients (fp32 HIP vs fp64 oracle)


@pytest.fixture(scope='module')
def tab():
    return nsgt.nsgt_tables(9, 60, SR, N)


@pytest.fixture(scope='module')
def cqt():
    from timbre_trap.framework import CQT
    return CQT(9, 60, SR, 3).to('cuda')


def _audio(B, nblk, seed=0):
    g = torch.Generator().manual_seed(seed)
    a = torch.rand(B, 1, nblk * N, generator=g) * 2 - 1
    return a / a.abs().amax(dim=-1, keepdim=True)


def test_forward_matches_oracle(cqt, tab):
    a = _audio(2, 2)
    out = cqt(a.cuda())
    assert out.shape == (2, 2, 540, 2 * M) and out.dtype == torch.float32 and out.is_contiguous()
    ref = nsgt.wrapper_forward(a.numpy(), tab)
    err = np.abs(out.cpu().numpy() - ref).max() / np.abs(ref).max()
    assert err < REL, err
    # per-bin relative error as well (low bins have small magnitudes)
    o = out.cpu().numpy()
    per_bin = np.abs(o - ref).max(axis=(0, 1, 3)) / np.abs(ref).max(axis=(0, 1, 3))
    assert per_bin.max() < 2e-3, per_bin.argmax()


def test_encode_is_complex_view_of_forward(cqt):
    a = _audio(1, 1, seed=3).cuda()
    c = cqt.encode(a)
    assert c.shape == (1, 1, 540, M) and c.is_complex()
    r = cqt(a)
    # same arithmetic, two template instantiations of the kernel: equal up to fused-multiply-add contraction
    tol = 1e-6 * float(r.abs().max())
    assert (cqt.to_real(c).contiguous() - r).abs().max() <= tol
    assert torch.equal(cqt.to_complex(cqt.to_real(c).contiguous()), c[:, 0])      # the layout helpers are exact inverses


def test_blocks_independent_and_linear(cqt):
    a1, a2 = _audio(1, 1, 5).cuda(), _audio(1, 1, 6).cuda()
    c1, c2 = cqt(a1), cqt(a2)
    c12 = cqt(torch.cat([a1, a2], -1))
    assert torch.equal(c12, torch.cat([c1, c2], -1))                      # bit-exact frame concatenation
    lin = cqt(0.5 * a1 - 2.0 * a2)
    scale = c1.abs().max()
    assert ((lin - (0.5 * c1 - 2.0 * c2)).abs().max() / scale) < 1e-5
    both = cqt(torch.cat([a1, a2], 0))
    assert torch.equal(both[0:1], c1) and torch.equal(both[1:2], c2)       # batch items independent


def test_sinusoid_and_click(cqt, tab):
    k = 333
    f = tab['positions'][k] * SR / N
    x = torch.cos(2 * np.pi * f * torch.arange(N, dtype=torch.float64) / SR).float().view(1, 1, N)
    mag = cqt.to_magnitude(cqt(x.cuda()))[0]
    assert int(mag.mean(-1).argmax()) == k
    x = torch.zeros(1, 1, N)
    x[0, 0, 33075] = 1.0
    mag = cqt.to_magnitude(cqt(x.cuda()))[0, 300:]
    assert abs(int(mag.sum(0).argmax()) - 33075 / (N / M)) <= 1.0


def test_inverse_matches_oracle(cqt, tab):
    a = _audio(2, 2, seed=7)
    ref_c = nsgt.wrapper_forward(a.numpy(), tab)
    c = torch.from_numpy(np.ascontiguousarray(ref_c)).float().cuda()
    out = cqt.decode(c)
    assert out.shape == (2, 1, 2 * N)
    ref = nsgt.wrapper_decode(ref_c, tab)
    assert np.abs(out.cpu().numpy() - ref).max() < REL
    assert abs(float(out.abs().max()) - 1.0) < 1e-6                          # inf-norm of the whole batch tensor
    # complex (B,1,F,T) input is accepted too (cqtwrapper.py:199-203)
    cc = torch.complex(c[:, 0], c[:, 1]).unsqueeze(1)
    assert torch.allclose(cqt.decode(cc), out, atol=1e-6)
    # arbitrary (non-consistent) coefficients, e.g. network outputs
    g = torch.Generator().manual_seed(11)
    rnd = torch.randn(1, 2, 540, M, generator=g)
    ref = nsgt.wrapper_decode(rnd.numpy().astype(np.float64), tab)
    assert np.abs(cqt.decode(rnd.cuda()).cpu().numpy() - ref).max() < REL


def test_round_trip_on_covered_band(cqt, tab):
    rng = np.random.default_rng(2)
    x = rng.uniform(-1, 1, (3, 1, N))
    X = np.fft.rfft(x, axis=-1)
    X[..., tab['frame_diag'] <= 1e-3] = 0        # the well-covered band (the default additive dual inverts it to 1e-8 / D per index)
    xb = np.fft.irfft(X, n=N, axis=-1)
    xb = xb / np.abs(xb).max()
    a = torch.from_numpy(xb).float().cuda()
    y = cqt.decode(cqt(a))
    assert (y - a).abs().max() < REL


def test_zeros_and_errors(cqt):
    y = cqt.decode(torch.zeros(1, 2, 540, M, device='cuda'))
    assert y.shape == (1, 1, N) and float(y.abs().max()) == 0.0
    with pytest.raises(ValueError):
        cqt(torch.zeros(1, 1, N + 5, device='cuda'))
    with pytest.raises(RuntimeError):
        cqt(torch.zeros(1, 1, N))                                           # CPU tensor: no fallback
    padded = cqt.pad_to_block_length(torch.zeros(2, 1, int(2.5 * N), device='cuda'))
    assert cqt(padded).shape == (2, 2, 540, 3 * M)


CONVENTION_SETS = {
    'floored_dual': dict(dual='floored'),
    'symmetric_canonical': dict(window='hann_symmetric', dual='canonical'),
    'floor_ceil': dict(length_rounding='floor', centre_rounding='ceil', min_length=2),
    'window_start': dict(crop_alignment='window_start', length_rounding='ceil'),
}


@pytest.mark.parametrize('name', list(CONVENTION_SETS))
def test_other_conventions_are_only_tables(name):
    """
    Every convention switch (window family, rounding rules, crop alignment, dual rule) reaches the kernels as DATA: the same
    HIP code run on another plan agrees with the independently written dense oracle at that point of the convention space.
    """
    from oracle.nsgt_dense import DenseNSGT
    from timbre_trap.framework import CQT
    from timbre_trap.framework.nsgt_plan import NSGTConventions
    kw = CONVENTION_SETS[name]
    cq = CQT(9, 60, 22050, 3, conventions=NSGTConventions(**kw)).cuda()
    d = DenseNSGT(9, 60, 22050, 66150, conventions=kw)
    g = torch.Generator().manual_seed(3)
    audio = (torch.rand(2, 1, 66150, generator=g) * 2 - 1)
    want = d.encode(audio.numpy().astype(np.float64))
    got = cq.encode(audio.cuda()).cpu().numpy()
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 1e-4 * np.abs(want).max()
    if kw.get('dual') == 'canonical':
        return      # fp32 round-off times dual gains of up to ~6e4 at the band edges: not comparable at a fixed tolerance (why the floor exists)
    # inverse: synthesise a spectrum-covering coefficient set (the analysis of noise) and compare the normalised audio
    back = cq.decode(torch.from_numpy(want.astype(np.complex64)).cuda()).cpu().numpy()
    ref = d.decode(want)
    ref = ref / np.abs(ref).max()
    assert np.abs(back - ref).max() < 1e-3


@pytest.mark.parametrize('B,nblk', [(16, 1), (19, 1), (9, 2), (5, 3)])
def test_batch_is_bit_identical_to_single_clips(cqt, B, nblk):
    """Every clip of a batch equals the clip transformed alone, forward and inverse (inverse without the batch-wide normalisation,
    which depends on the other clips by design), whatever the launch geometry: the property a split of the batch into chunks must
    keep (round 4 measured a two-stream chunk pipeline inside tt_cqt_forward / tt_cqt_inverse: bit-identical, but every
    cross-stream event costs ~13 us on this stack -- profiles/r04_cqt_stream_split.txt -- so it was dropped)."""
    a = _audio(B, nblk, seed=21).cuda()
    c = cqt(a)
    for b in range(B):
        assert torch.equal(c[b:b + 1], cqt(a[b:b + 1])), b
    raw = cqt._decode_raw(c)
    back = cqt.decode(c)
    torch.cuda.synchronize()
    assert back.shape == a.shape and bool(torch.isfinite(back).all())
    assert abs(float(back.abs().max()) - 1.0) < 1e-6
    assert torch.equal(back, cqt.decode(c))                      # the abs-max is an atomicMax over the batch: order-free
    for b in range(B):
        assert torch.equal(raw[b:b + 1], cqt._decode_raw(c[b:b + 1])), b


# ---- any block length (csrc/cqt_generic.hip; reference cqtwrapper.py:15-48 takes arbitrary secs_per_block / sample_rate) ------------

def _generic_cqt(monkeypatch, secs):
    from timbre_trap.framework import CQT, cqtwrapper
    monkeypatch.setattr(cqtwrapper, 'FORCE_GENERIC', True)
    return CQT(9, 60, SR, secs).to('cuda')


def test_generic_path_equals_specialised_path_at_the_reference_configuration(cqt, tab, monkeypatch):
    """N = 66150 / M = 1024 forced onto the any-length kernels (Bluestein + global Stockham passes): the same tables, so the same
    transform as the specialised kernels -- forward against the oracle at the north-star bar and against the fast path at fp32
    level, inverse likewise, complex in / out, batch-wide normalisation, the zero guard."""
    g = _generic_cqt(monkeypatch, 3)
    assert not g._fast and cqt._fast and g.block_length == N and g.max_window_length == M
    a = _audio(3, 2, seed=31)
    ref = nsgt.wrapper_forward(a.numpy(), tab)
    fast, slow = cqt(a.cuda()), g(a.cuda())
    scale = np.abs(ref).max()
    assert slow.shape == fast.shape and slow.is_contiguous()
    assert np.abs(slow.cpu().numpy() - ref).max() / scale < REL
    assert float((slow - fast).abs().max()) / scale < 2e-5
    ce = g.encode(a.cuda())
    assert ce.is_complex() and ce.shape == (3, 1, 540, 2 * M)
    assert torch.equal(g.to_real(ce).contiguous(), slow)
    back_f, back_s = cqt.decode(fast), g.decode(fast)
    want = nsgt.wrapper_decode(fast.cpu().numpy().astype(np.float64), tab)
    assert np.abs(back_s.cpu().numpy() - want).max() < REL and float((back_s - back_f).abs().max()) < REL
    assert abs(float(back_s.abs().max()) - 1.0) < 1e-6
    assert torch.allclose(g.decode(torch.complex(fast[:, 0], fast[:, 1]).unsqueeze(1)), back_s, atol=1e-6)
    raw = g._decode_raw(fast)
    assert torch.equal(raw[1:2], g._decode_raw(fast[1:2]))                   # clips independent before the normalisation
    z = g.decode(torch.zeros(1, 2, 540, M, device='cuda'))
    assert z.shape == (1, 1, N) and float(z.abs().max()) == 0.0


@pytest.mark.parametrize('secs', [2, 4, 2.9, 0.5])
def test_other_block_lengths(secs):
    """CQT(9, 60, 22050, secs_per_block != 3): 44100 (M = 512), 88200 (M = 1024), 63945 = 3^2 5 7^2 29 (a prime factor no
    mixed-radix plan would cover) and 11025 samples (M = 128) -- constructor attributes as the reference computes them
    (cqtwrapper.py:40-48), forward / inverse against the float64 oracle built for that block length, block independence."""
    from timbre_trap.framework import CQT
    n = int(secs * SR)
    t = nsgt.nsgt_tables(9, 60, SR, n)
    m = t['max_window_length']
    cq = CQT(9, 60, SR, secs).to('cuda')
    assert cq.block_length == n and cq.max_window_length == m and cq.hop_length == n / m and cq.n_bins == 540
    assert cq.get_expected_frames(2 * n) == 2 * m
    g = torch.Generator().manual_seed(int(secs * 10))
    a = torch.rand(2, 1, 2 * n, generator=g) * 2 - 1
    out = cq(a.cuda())
    assert out.shape == (2, 2, 540, 2 * m)
    ref = nsgt.wrapper_forward(a.numpy(), t)
    assert np.abs(out.cpu().numpy() - ref).max() / np.abs(ref).max() < REL
    assert torch.equal(out[..., :m], cq(a[..., :n].cuda())) and torch.equal(out[1:2], cq(a[1:2].cuda()))
    back = cq.decode(out)
    want = nsgt.wrapper_decode(out.cpu().numpy().astype(np.float64), t)
    assert back.shape == (2, 1, 2 * n) and np.abs(back.cpu().numpy() - want).max() < REL
    with pytest.raises(ValueError):
        cq(torch.zeros(1, 1, n + 1, device='cuda'))
