"""
ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement, in numpy float64, of the invertible constant-Q transform that
``timbre_trap.framework.CQT`` (reference ``timbre_trap/framework/cqtwrapper.py``)
obtains from the third-party package ``cqt_pytorch`` (archinetai/cqt-pytorch,
UNPINNED in reference ``requirements.txt:15``, absent from /root/reference and from
this image).

    *** PARITY UNPINNED ***
    The reference holds no test, golden vector or fixture for the CQT values and
    the package that computes them cannot be imported here.  What IS pinned, by the
    reference's own arithmetic, is restated and checked in tests/:
      - 540 bins, geometric from fmin = (sr/2)/2**n_octaves   (cqtwrapper.py:45-48)
      - every bin yields max_window_length = 1024 frames per 66150-sample block,
        a power of two                                         (cqtwrapper.py:35,40,271)
      - blocks are independent, frames concatenate             (cqtwrapper.py:231,271)
      - encode -> complex (B,1,F,T); to_real -> (B,2,F,T)      (cqtwrapper.py:67-95)
      - decode(encode(x)) ~ x up to scale, then inf-norm       (cqtwrapper.py:207-211)
    The remaining conventions (window family / length rounding / centre rounding /
    crop alignment / dual window) follow the published NSGT construction
    (Velasco, Holighaus, Doerfler, Grill 2011; Holighaus et al. 2013, "painless"
    case) with the parameter choices recalled from cqt_pytorch; every one of them
    lives in :func:`nsgt_tables`, and both this oracle and the HIP path consume
    nothing but those tables, so a different convention is a different table.

Algorithm (one block of N samples, M = max_window_length):
    X      = fft_N(x)
    v_k[m] = X[(start_k + m) mod N] * g_k[m]           m in [0, M), g_k zero outside
                                                       [pad_k, pad_k + L_k)
    c_k    = ifft_M(v_k)                               (numpy/torch 1/M convention)
    inverse:
    V_k    = fft_M(c_k)
    Xh[j]  = sum_k V_k[j - start_k] * gd_k[j - start_k]        gd_k = g_k / D,
             D[j] = sum_k g_k[j - start_k]**2   (diagonal frame operator)
    xh     = real(ifft_N(hermitian(Xh)))
"""

import math

import numpy as np


FRAME_FLOOR = 1e-3      # the opt-in 'floored' dual (rounds 1-4 default)
DUAL_EPS = 1e-8         # the default 'additive' dual: g / (D + eps), the recalled dense one-liner of cqt_pytorch (unverified)


def _round_half_even(x):
    # torch.round / numpy.round semantics (banker's rounding), as a tensor op in cqt_pytorch
    return np.round(np.asarray(x, dtype=np.float64))


def nsgt_tables(n_octaves, bins_per_octave, sample_rate, block_length, power_of_2_length=True, dual='additive'):
    """
    All conventions of the transform, as plain tables.  ``dual``: 'additive' (default, g / (D + 1e-8) wherever a window reaches) |
    'floored' (g / D where D > 1e-3, else 0) | 'canonical' (g / D wherever D > 0).

    Returns a dict with
      n_bins, block_length (N), max_window_length (M)
      freqs[k]        centre frequency in Hz
      lengths[k]      L_k  window length in spectral samples (>= 1)
      positions[k]    c_k  centre index in the length-N spectrum
      start[k]        first spectral index of the M-long crop (may be negative -> mod N)
      pad[k]          offset of the window inside the crop
      win_off[k]      prefix offsets into the ragged arrays (win_off[-1] = sum L_k)
      window          ragged analysis windows  g_k  (float64, sum L_k)
      dual            ragged synthesis windows gd_k (float64, sum L_k)
      spec_index      ragged absolute spectral index of every window sample (int64)
      covered         boolean (N//2+1) - spectral indices with D[j] > 0
    """
    n_bins = n_octaves * bins_per_octave
    N = int(block_length)

    f_nyq = sample_rate / 2
    f_min = f_nyq / (2 ** n_octaves)
    k = np.arange(n_bins, dtype=np.float64)
    freqs = f_min * 2.0 ** (k / bins_per_octave)

    # constant-Q bandwidth  Omega_k = f_k * (2^(1/B) - 2^(-1/B))
    q_inv = 2.0 ** (1.0 / bins_per_octave) - 2.0 ** (-1.0 / bins_per_octave)
    bandwidths = freqs * q_inv

    lengths = np.maximum(_round_half_even(bandwidths * N / sample_rate), 1).astype(np.int64)
    M = int(lengths.max())
    if power_of_2_length:
        M = 2 ** int(math.ceil(math.log2(M)))

    positions = _round_half_even(freqs * N / sample_rate).astype(np.int64)

    pad = np.floor(M / 2 - lengths / 2).astype(np.int64)
    start = positions - M // 2

    win_off = np.zeros(n_bins + 1, dtype=np.int64)
    win_off[1:] = np.cumsum(lengths)
    total = int(win_off[-1])

    window = np.zeros(total, dtype=np.float64)
    spec_index = np.zeros(total, dtype=np.int64)
    for b in range(n_bins):
        L = int(lengths[b])
        n = np.arange(L, dtype=np.float64)
        # periodic Hann (torch.hann_window default); length-1 window is [1.]
        g = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / L) if L > 1 else np.ones(1)
        window[win_off[b]:win_off[b + 1]] = g
        spec_index[win_off[b]:win_off[b + 1]] = start[b] + pad[b] + np.arange(L)

    assert spec_index.min() > 0 and spec_index.max() < N // 2, \
        'a window leaves the open positive half-spectrum; hermitian wrap not restated'

    # diagonal of the frame operator on the positive half-spectrum
    D = np.zeros(N // 2 + 1, dtype=np.float64)
    np.add.at(D, spec_index, window ** 2)
    # 'floored': spectral indices whose total window energy is below FRAME_FLOOR are treated as not represented: at the two
    # band edges (and two sub-43 Hz gaps) only the extreme tail of ONE window reaches them, the canonical dual there would be
    # 1/w ~ 6e4 and turn 1e-4 coefficient noise into audible sinusoids.  'additive' (default since round 5: the builder's best
    # recollection of cqt_pytorch) regularises with an epsilon in the denominator instead.
    if dual == 'additive':
        covered = D > 0.0
        dual = np.where(covered[spec_index], window / (D[spec_index] + DUAL_EPS), 0.0)
    elif dual in ('floored', 'canonical'):
        covered = D > (FRAME_FLOOR if dual == 'floored' else 0.0)
        Dsafe = np.where(covered, D, 1.0)
        dual = np.where(covered[spec_index], window / Dsafe[spec_index], 0.0)
    else:
        raise ValueError('unknown dual rule %r' % (dual,))

    return dict(n_bins=n_bins, block_length=N, max_window_length=M, freqs=freqs,
                lengths=lengths, positions=positions, start=start, pad=pad,
                win_off=win_off, window=window, dual=dual, spec_index=spec_index,
                covered=covered, frame_diag=D)


def encode(audio, tab):
    """
    NSGT analysis.  audio: (B, 1, n*N) real -> complex128 (B, 1, F, n*M).
    Mirrors the call ``self.encode(audio)`` at cqtwrapper.py:67.
    """
    audio = np.asarray(audio, dtype=np.float64)
    N, M, F = tab['block_length'], tab['max_window_length'], tab['n_bins']
    B = audio.shape[0]
    assert audio.shape[-1] % N == 0, 'pad to a multiple of the block length first'
    nblk = audio.shape[-1] // N
    x = audio.reshape(B, nblk, N)
    X = np.fft.fft(x, axis=-1)

    out = np.zeros((B, nblk, F, M), dtype=np.complex128)
    off = tab['win_off']
    for k in range(F):
        L = off[k + 1] - off[k]
        v = np.zeros((B, nblk, M), dtype=np.complex128)
        idx = tab['spec_index'][off[k]:off[k + 1]] % N
        v[..., tab['pad'][k]:tab['pad'][k] + L] = X[..., idx] * tab['window'][off[k]:off[k + 1]]
        out[:, :, k, :] = np.fft.ifft(v, axis=-1)
    # (B, n, F, M) -> (B, 1, F, n*M): frames of successive blocks concatenate in time
    return out.transpose(0, 2, 1, 3).reshape(B, 1, F, nblk * M)


def decode(coefficients, tab):
    """
    NSGT synthesis (no inf-norm).  complex (B, 1, F, n*M) -> real (B, 1, n*N).
    Mirrors ``super().decode(coefficients)`` at cqtwrapper.py:207.
    """
    c = np.asarray(coefficients)
    N, M, F = tab['block_length'], tab['max_window_length'], tab['n_bins']
    B = c.shape[0]
    nblk = c.shape[-1] // M
    c = c.reshape(B, F, nblk, M)
    V = np.fft.fft(c, axis=-1)

    Xh = np.zeros((B, nblk, N), dtype=np.complex128)
    off = tab['win_off']
    for k in range(F):
        L = off[k + 1] - off[k]
        idx = tab['spec_index'][off[k]:off[k + 1]]
        seg = V[:, k, :, tab['pad'][k]:tab['pad'][k] + L] * tab['dual'][off[k]:off[k + 1]]
        Xh[..., idx] += seg
    # hermitian extension: the bins only span the open positive half-spectrum
    j = np.arange(1, (N + 1) // 2)
    Xh[..., N - j] = np.conj(Xh[..., j])
    x = np.fft.ifft(Xh, axis=-1).real
    return x.reshape(B, 1, nblk * N)


# ---- wrapper behaviour restated from cqtwrapper.py (all lines cited) -------------------------

def to_real(c):
    """cqtwrapper.py:74-97 : complex (B,1,F,T) -> (B,2,F,T), ch0 = re, ch1 = im."""
    c = np.asarray(c)[:, 0]
    return np.stack([c.real, c.imag], axis=1)


def to_complex(r):
    """cqtwrapper.py:99-120 : (B,2,F,T) -> complex (B,F,T)."""
    r = np.asarray(r)
    return r[:, 0] + 1j * r[:, 1]


def to_magnitude(r):
    """cqtwrapper.py:122-141 : L2 norm over the channel dim."""
    r = np.asarray(r)
    return np.sqrt((r ** 2).sum(axis=-3))


def to_decibels(mag, rescale=True):
    """
    cqtwrapper.py:143-182 with torchaudio AmplitudeToDB('amplitude', top_db=80):
    db = 20*log10(clamp(m, 1e-10)) - 20*log10(max(1e-10, 1.0)); clamp at max-80 per item.
    """
    mag = np.asarray(mag, dtype=np.float64)
    out = []
    for m in mag:
        d = 20.0 * np.log10(np.maximum(m, 1e-10))
        d = np.maximum(d, d.max() - 80.0)
        if rescale:
            d = d - d.max()
            d = 1 + d / 80
        out.append(d)
    return np.stack(out)


def wrapper_forward(audio, tab):
    """cqtwrapper.py:50-72."""
    return to_real(encode(audio, tab))


def wrapper_decode(coefficients, tab):
    """cqtwrapper.py:184-213 : real (B,2,F,T) or complex (B,1,F,T) -> inf-normalised audio."""
    c = np.asarray(coefficients)
    if not np.iscomplexobj(c):
        c = to_complex(c)[:, None]
    audio = decode(c, tab)
    peak = np.abs(audio).max()
    if peak:
        audio = audio / peak
    return audio


def pad_to_block_length(audio, N):
    """cqtwrapper.py:215-233."""
    audio = np.asarray(audio)
    p = -audio.shape[-1] % N
    return np.pad(audio, [(0, 0)] * (audio.ndim - 1) + [(0, p)])


def get_expected_samples(t, sample_rate):
    """cqtwrapper.py:235-253."""
    return int(max(0, t) * sample_rate)


def get_expected_frames(num_samples, N, M):
    """cqtwrapper.py:255-273."""
    return math.ceil((num_samples / N) * M)


def get_times(n_frames, N, M, sample_rate):
    """cqtwrapper.py:275-293 with hop_length = N / M (cqtwrapper.py:40)."""
    return np.arange(n_frames) * (N / M) / sample_rate


def get_midi_freqs(n_octaves, bins_per_octave, sample_rate):
    """cqtwrapper.py:43-48; librosa.hz_to_midi(f) = 12*(log2(f) - log2(440)) + 69."""
    fmin = 12 * (np.log2(np.asanyarray((sample_rate / 2) / (2 ** n_octaves))) - np.log2(440.0)) + 69
    return fmin + np.arange(n_octaves * bins_per_octave) / (bins_per_octave / 12)
