"""
Cross-check of the three statements of the NSGT tables / transform (CPU): the product's plan builder
(timbre_trap/framework/nsgt_plan.py, ragged tables), the ragged oracle (oracle/nsgt.py) and the independently written dense
oracle (oracle/nsgt_dense.py: closed-form windows broadcast over the absolute spectral index, transform as index-grid gathers),
for the default conventions and for other points of the convention space (every switch of NSGTConventions flipped at least
once).  All three remain PARITY UNPINNED against cqt_pytorch -- this pins them against each other.
"""

import numpy as np
import pytest

from oracle import nsgt
from oracle.nsgt_dense import DenseNSGT
from timbre_trap.framework import nsgt_plan

N, SR = 66150, 22050
CONVENTION_SETS = {
    'default': {},
    'symmetric_canonical': dict(window='hann_symmetric', dual='canonical'),
    'floor_ceil': dict(length_rounding='floor', centre_rounding='ceil', min_length=2),
    'window_start': dict(crop_alignment='window_start', length_rounding='ceil'),
    'additive_dual': dict(dual='additive', dual_eps=1e-6),
}


def _plan_as_tab(plan):
    """The product's plan in the key names oracle/nsgt.py's encode / decode read."""
    return dict(n_bins=plan['n_bins'], block_length=plan['N'], max_window_length=plan['M'], win_off=plan['win_off'], pad=plan['pad'],
                spec_index=plan['spec_index'], window=plan['window'], dual=plan['dual'])


@pytest.mark.parametrize('name', list(CONVENTION_SETS))
def test_plan_tables_equal_the_dense_derivation(name):
    kw = CONVENTION_SETS[name]
    plan = nsgt_plan.build_plan(9, 60, SR, N, conventions=nsgt_plan.NSGTConventions(**kw))
    d = DenseNSGT(9, 60, SR, N, conventions=kw)
    g = d.geo
    assert plan['M'] == g['M'] == 1024
    np.testing.assert_array_equal(plan['lengths'], g['L'])
    np.testing.assert_array_equal(plan['positions'], g['c'])
    np.testing.assert_array_equal(plan['start'], g['s'])
    np.testing.assert_array_equal(plan['start'] + plan['pad'], g['a'])
    off = plan['win_off']
    for k in (0, 1, 44, 45, 100, 270, 400, 538, 539):
        seg = slice(off[k], off[k + 1])
        j = plan['spec_index'][seg]
        np.testing.assert_allclose(plan['window'][seg], d.W[k, j], rtol=0, atol=1e-15)
        np.testing.assert_allclose(plan['dual'][seg], d.Wd[k, j], rtol=1e-13, atol=0)
        assert d.W[k].sum() == pytest.approx(plan['window'][seg].sum(), rel=1e-14)       # nothing of the row outside the segment
    np.testing.assert_allclose(plan['frame_diag'], d.D, rtol=1e-13, atol=1e-300)
    np.testing.assert_array_equal(plan['covered'], d.kept)


@pytest.mark.parametrize('name', list(CONVENTION_SETS))
def test_ragged_and_dense_transforms_agree(name):
    kw = CONVENTION_SETS[name]
    plan = nsgt_plan.build_plan(9, 60, SR, N, conventions=nsgt_plan.NSGTConventions(**kw))
    tab = _plan_as_tab(plan)
    d = DenseNSGT(9, 60, SR, N, conventions=kw)
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, (1, 1, 2 * N))
    c_r, c_d = nsgt.encode(x, tab), d.encode(x)
    assert c_r.shape == c_d.shape == (1, 1, 540, 2048)
    assert np.abs(c_r - c_d).max() <= 1e-12 * np.abs(c_d).max()
    a_r, a_d = nsgt.decode(c_r, tab), d.decode(c_d)
    assert np.abs(a_r - a_d).max() <= 1e-12 * np.abs(a_d).max()


def test_default_oracle_tables_are_the_default_conventions():
    """oracle/nsgt.py (fixed conventions) == dense derivation at its defaults: the pre-existing oracle is one point of the space."""
    tab = nsgt.nsgt_tables(9, 60, SR, N)
    d = DenseNSGT(9, 60, SR, N)
    np.testing.assert_array_equal(tab['start'] + tab['pad'], d.geo['a'])
    np.testing.assert_array_equal(tab['covered'], d.kept)
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, (1, 1, N))
    assert np.abs(nsgt.encode(x, tab) - d.encode(x)).max() < 1e-12


def test_canonical_dual_inverts_exactly_on_the_covered_band_and_floor_is_the_only_difference():
    """What the opt-in 'floored' rule changes: only spectral indices with frame-operator diagonal <= 1e-3 (a handful); and what the
    default 'additive' rule (g / (D + 1e-8)) changes against the canonical dual: a relative 1e-8 / D per index, i.e. nothing on
    the covered band and a bounded gain (<= 5e3 instead of ~6e4) where only a window tail reaches."""
    floored, canon = DenseNSGT(9, 60, SR, N, conventions=dict(dual='floored')), DenseNSGT(9, 60, SR, N, conventions=dict(dual='canonical'))
    additive = DenseNSGT(9, 60, SR, N)
    assert additive.geo['cv']['dual'] == 'additive' and np.array_equal(additive.kept, canon.kept)
    assert np.abs(additive.Wd).max() <= 0.5 / np.sqrt(1e-8) and np.abs(canon.Wd).max() > 1e4
    strong = canon.D > 1e-3
    assert np.abs(additive.Wd[:, strong] - canon.Wd[:, strong]).max() <= 1.01e-5 * np.abs(canon.Wd[:, strong]).max()
    differ = np.flatnonzero(floored.kept != canon.kept)
    assert 0 < len(differ) <= 64          # the upper tail of the top window (33030..33071) and a few sub-43 Hz indices
    assert canon.D[differ].max() <= 1e-3
    # a band-limited signal living strictly inside the floored band reconstructs identically under both rules
    rng = np.random.default_rng(2)
    X = np.zeros(N // 2 + 1, dtype=np.complex128)
    band = np.flatnonzero(floored.kept)
    band = band[(band > 200) & (band < 30000)]
    X[band] = rng.normal(size=len(band)) + 1j * rng.normal(size=len(band))
    x = np.fft.irfft(X, n=N)[None, None]
    for t, tol in ((floored, 1e-9), (canon, 1e-9), (additive, 1e-6)):
        back = t.decode(t.encode(x))
        assert np.abs(back - x).max() < tol * np.abs(x).max()


@pytest.mark.parametrize('n,m,p', [(44100, 512, 131072), (88200, 1024, 262144), (66150, 1024, 262144), (63945, 1024, 131072)])
def test_any_block_length_plan(n, m, p):
    """The tables of the any-length device path (csrc/cqt_generic.hip) for N in {66150, 44100, 88200} and a block length with a large
    prime factor: geometry against the dense derivation at that N, and the Bluestein tables as KNOWN-ANSWER data -- the length-N DFT
    computed from them exactly as the kernels do (chirp, two length-P FFTs around the filter table, chirp), in float32 tables,
    against numpy's FFT."""
    plan = nsgt_plan.build_plan(9, 60, SR, n, generic=True)
    assert not plan['fast'] and plan['M'] == m and plan['P'] == p and p >= 2 * n - 1
    d = DenseNSGT(9, 60, SR, n)
    np.testing.assert_array_equal(plan['lengths'], d.geo['L'])
    np.testing.assert_array_equal(plan['start'] + plan['pad'], d.geo['a'])
    np.testing.assert_allclose(plan['frame_diag'], d.D, rtol=1e-13, atol=1e-300)
    assert plan['pos_bin'].shape == (plan['sum_len'],) and plan['pos_bin'][plan['win_off'][7]] == 7
    c = lambda a: a[:, 0].astype(np.float32).astype(np.float64) + 1j * a[:, 1].astype(np.float32).astype(np.float64)
    chirp, bfilt = c(plan['chirp']), c(plan['bfilt'])
    x = np.random.default_rng(n).uniform(-1, 1, n)
    a = np.zeros(p, dtype=np.complex128)
    a[:n] = x * chirp
    w = np.fft.fft(np.conj(np.fft.fft(a) * bfilt))              # the inverse transform as conj(FFT(conj .)); 1 / P is in the table
    X = chirp * np.conj(w[:n])
    want = np.fft.fft(x)
    assert np.abs(X - want).max() < 2e-6 * np.abs(want).max()
    tw = c(plan['twM'])
    np.testing.assert_allclose(tw, np.exp(-2j * np.pi * np.arange(m // 2) / m), atol=1e-7)
