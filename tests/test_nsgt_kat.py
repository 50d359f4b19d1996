"""
Known-answer tests of the NSGT oracle (oracle/nsgt.py) -- the properties the reference's own
arithmetic pins at the cqt_pytorch boundary (SURVEY.md section 8c list).  CPU only.
The same properties are checked on the HIP path at full size in tests/test_gpu_cqt.py.
"""

import numpy as np
import pytest

from oracle import nsgt

N, M, SR = 66150, 1024, 22050


@pytest.fixture(scope='module')
def tab():
    return nsgt.nsgt_tables(9, 60, SR, N)


def test_geometry(tab):
    assert tab['n_bins'] == 540 and tab['max_window_length'] == 1024
    assert np.isclose(tab['freqs'][0], 11025 / 512) and np.isclose(tab['freqs'][-1], 11025 * 2 ** (-1 / 60))
    assert tab['lengths'].max() == 755 and tab['lengths'].min() == 1
    assert int(tab['win_off'][-1]) == 65649
    # every spectral index between the first and last centre is covered except two sub-43 Hz gaps
    c0, c1 = tab['positions'][0], tab['positions'][-1]
    gaps = np.nonzero(~tab['covered'][c0:c1 + 1])[0] + c0
    assert gaps.tolist() == [94, 102]


def test_shapes_blocks_linearity(tab):
    rng = np.random.default_rng(0)
    x1, x2 = rng.uniform(-1, 1, (2, 1, N)), rng.uniform(-1, 1, (2, 1, N))
    c1, c2 = nsgt.encode(x1, tab), nsgt.encode(x2, tab)
    assert c1.shape == (2, 1, 540, 1024)
    c12 = nsgt.encode(np.concatenate([x1, x2], -1), tab)
    assert np.array_equal(c12, np.concatenate([c1, c2], -1))              # independent blocks, frames concatenate
    np.testing.assert_allclose(nsgt.encode(0.5 * x1 - 2 * x2, tab), 0.5 * c1 - 2 * c2, atol=1e-9)
    r = nsgt.to_real(c1)
    assert r.shape == (2, 2, 540, 1024) and np.array_equal(nsgt.to_complex(r), c1[:, 0])


def test_sinusoid_peaks_at_its_bin(tab):
    for k in (200, 333, 480):
        f = tab['positions'][k] * SR / N                                     # exactly on the bin's centre index
        x = np.cos(2 * np.pi * f * np.arange(N) / SR)[None, None]
        mag = np.abs(nsgt.encode(x, tab))[0, 0]
        assert mag.mean(-1).argmax() == k
        assert mag[k].std() / mag[k].mean() < 1e-6                           # stationary over frames
        midi = nsgt.get_midi_freqs(9, 60, SR)[k]
        assert abs(midi - (12 * np.log2(tab['freqs'][k] / 440) + 69)) < 1e-9


def test_click_peaks_at_its_frame(tab):
    for n0 in (10000, 33075, 50001):
        x = np.zeros((1, 1, N))
        x[0, 0, n0] = 1.0
        mag = np.abs(nsgt.encode(x, tab))[0, 0, 300:]                        # bins with useful time resolution
        frame = mag.sum(0).argmax()
        assert abs(frame - n0 / (N / M)) <= 1.0


@pytest.mark.parametrize('dual,tol', [('additive', 1e-6), ('canonical', 1e-10), ('floored', 1e-10)])
def test_perfect_reconstruction_on_covered_band(dual, tol):
    """decode(encode(x)) = x for signals on the well-covered band (frame-operator diagonal > 1e-3): exactly under the canonical and
    the floored dual, to a relative 1e-8 / D per spectral index under the default additive one (g / (D + 1e-8))."""
    tab = nsgt.nsgt_tables(9, 60, SR, N, dual=dual)
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, (2, 2, N))
    X = np.fft.rfft(x, axis=-1)
    X[..., tab['frame_diag'] <= 1e-3] = 0
    xb = np.fft.irfft(X, n=N, axis=-1).reshape(2, 1, 2 * N)
    y = nsgt.decode(nsgt.encode(xb, tab), tab)
    assert np.abs(y - xb).max() < tol
    yn = nsgt.wrapper_decode(nsgt.wrapper_forward(xb, tab), tab)
    np.testing.assert_allclose(yn * np.abs(xb).max(), xb, atol=tol)
    assert np.abs(yn).max() == 1.0


def test_decode_zeros_is_finite(tab):
    y = nsgt.wrapper_decode(np.zeros((1, 2, 540, 1024)), tab)
    assert y.shape == (1, 1, N) and not np.isnan(y).any() and np.all(y == 0)
