"""
ORACLE (test infrastructure only -- never imported by the product path).   *** PARITY UNPINNED (see oracle/nsgt.py) ***

A SECOND, independently written statement of the NSGT constant-Q transform, used to cross-check oracle/nsgt.py and the
product's plan builder (timbre_trap/framework/nsgt_plan.py), which were written together and share their ragged-table
structure.  This one has no ragged arrays, prefix offsets, CSR lists or per-bin loops: every bin's window is a dense row
over the absolute spectral index, built by broadcasting the closed-form window function over an (F x N/2+1) grid, and the
transform is written as gathers over an (F x M) index grid straight from the defining sums

    analysis   c_k[n]  = (1/M) sum_{m=0}^{M-1} X[s_k + m] W[k, s_k + m] e^{+2 pi i m n / M}          X = DFT_N(x)
    synthesis  Xh[j]   = sum_k  DFT_M(c_k)[j - s_k] Wd[k, j],     Wd = W / D,   D[j] = sum_k W[k, j]^2   (where D is kept)
               x       = Re IDFT_N(hermitian extension of Xh)

with  L_k = rnd(Omega_k N / sr),  c_k = rnd(f_k N / sr),  f_k = (sr/2) 2^(k/B - n_oct),  Omega_k = f_k (2^(1/B) - 2^(-1/B)).
The conventions (rounding rules, window family, where the M-sample crop s_k and the window's first index a_k sit, the floor
on D) arrive as a plain dict with the same keys as the product's NSGTConventions -- they are the free parameters of the
construction, shared by necessity; what is independent is every line that turns them into numbers.
"""

import math

import numpy as np

DEFAULTS = dict(window='hann_periodic', length_rounding='round', centre_rounding='round', crop_alignment='centred',
                dual='additive', frame_floor=1e-3, min_length=1, dual_eps=1e-8, bandwidth_bin=0, diagonal='positive')


def _rnd(x, rule):
    return {'round': np.rint, 'floor': np.floor, 'ceil': np.ceil}[rule](x)


def geometry(n_octaves, bins_per_octave, sample_rate, N, conventions=None, power_of_2_length=True):
    cv = dict(DEFAULTS, **(conventions or {}))
    F = n_octaves * bins_per_octave
    k = np.arange(F, dtype=np.float64)
    f = (sample_rate / 2.0) * np.exp2(k / bins_per_octave - n_octaves)
    # bandwidth_bin = 1 (hypothesis): the bandwidth of bin k taken from the NEXT bin's frequency
    omega = ((sample_rate / 2.0) * np.exp2((k + cv['bandwidth_bin']) / bins_per_octave - n_octaves)
             * (np.exp2(1.0 / bins_per_octave) - np.exp2(-1.0 / bins_per_octave)))
    L = np.maximum(_rnd(omega * N / sample_rate, cv['length_rounding']), cv['min_length']).astype(np.int64)
    M = int(L.max())
    if power_of_2_length:
        M = 1 << (M - 1).bit_length()
    c = _rnd(f * N / sample_rate, cv['centre_rounding']).astype(np.int64)
    if cv['crop_alignment'] == 'centred':
        s = c - M // 2                                   # first spectral index of the crop
        a = s + np.floor(M / 2.0 - L / 2.0).astype(np.int64)     # first index of the window: centred in the crop
    elif cv['crop_alignment'] == 'centre_minus_M':      # hypothesis: the crop starts a full M below the centre
        s = c - M
        a = s + np.floor(M / 2.0 - L / 2.0).astype(np.int64)
    else:
        a = c - L // 2
        s = a.copy()
    return dict(F=F, N=N, M=M, f=f, L=L, c=c, s=s, a=a, cv=cv)


def dense_windows(geo):
    """W (F x N/2+1): analysis window of every bin on the absolute spectral index; Wd: synthesis windows; D; kept."""
    F, N, L, a, cv = geo['F'], geo['N'], geo['L'], geo['a'], geo['cv']
    j = np.arange(N // 2 + 1, dtype=np.int64)[None, :]
    n = j - a[:, None]                                    # position inside the window, any integer
    inside = (n >= 0) & (n < L[:, None])
    Lf = L[:, None].astype(np.float64)
    denom = Lf if cv['window'] == 'hann_periodic' else np.maximum(Lf - 1.0, 1.0)
    W = np.where(inside, 0.5 - 0.5 * np.cos(2.0 * np.pi * n / denom), 0.0)
    W = np.where(inside & (L[:, None] == 1), 1.0, W)      # a one-sample window is [1]
    if (a <= 0).any() or (a + L > N // 2).any():
        raise ValueError('a window leaves the open positive half-spectrum')
    D = (W ** 2).sum(axis=0)
    # diagonal = 'mirrored' (hypothesis: the sum also runs over the mirror image of every window, at N - j): the images live on
    # N/2 < N - j < N -- outside this half-spectrum, since the check above keeps every window inside 1 .. N/2 - 1 -- and add nothing
    if cv['diagonal'] not in ('positive', 'mirrored'):
        raise ValueError('unknown frame-operator diagonal %r' % (cv['diagonal'],))
    kept = D > (cv['frame_floor'] if cv['dual'] == 'floored' else 0.0)
    if cv['dual'] == 'additive':                            # regularised inverse: every reached index, D + eps in the denominator
        Wd = np.where(kept[None, :], W / (D + cv['dual_eps'])[None, :], 0.0)
    else:
        Wd = np.where(kept[None, :], W / np.where(kept, D, 1.0)[None, :], 0.0)
    return W, Wd, D, kept


class DenseNSGT:
    def __init__(self, n_octaves=9, bins_per_octave=60, sample_rate=22050, block_length=66150, conventions=None):
        self.geo = geometry(n_octaves, bins_per_octave, sample_rate, int(block_length), conventions)
        self.W, self.Wd, self.D, self.kept = dense_windows(self.geo)
        g = self.geo
        self.idx = g['s'][:, None] + np.arange(g['M'], dtype=np.int64)[None, :]          # (F, M) absolute spectral indices
        self.valid = (self.idx >= 0) & (self.idx <= g['N'] // 2)
        self.idx_c = np.clip(self.idx, 0, g['N'] // 2)
        rows = np.arange(g['F'])[:, None]
        self.Wg = np.where(self.valid, self.W[rows, self.idx_c], 0.0)                     # windows gathered on the crop grid
        self.Wdg = np.where(self.valid, self.Wd[rows, self.idx_c], 0.0)

    def encode(self, audio):
        """(B, 1, n*N) real -> complex128 (B, 1, F, n*M)."""
        g = self.geo
        x = np.asarray(audio, dtype=np.float64)
        B = x.shape[0]
        nblk = x.shape[-1] // g['N']
        X = np.fft.rfft(x.reshape(B, nblk, g['N']), axis=-1)                              # (B, n, N/2+1)
        V = X[:, :, self.idx_c] * self.Wg                                                 # (B, n, F, M)
        c = np.fft.ifft(V, axis=-1)
        return c.transpose(0, 2, 1, 3).reshape(B, 1, g['F'], nblk * g['M'])

    def decode(self, coefficients):
        """complex (B, 1, F, n*M) -> real (B, 1, n*N), no normalisation."""
        g = self.geo
        c = np.asarray(coefficients)
        B = c.shape[0]
        nblk = c.shape[-1] // g['M']
        V = np.fft.fft(c.reshape(B, g['F'], nblk, g['M']), axis=-1) * self.Wdg[None, :, None, :]
        Xh = np.zeros((B, nblk, g['N'] // 2 + 1), dtype=np.complex128)
        flat_idx = self.idx_c.reshape(-1)
        contrib = V.transpose(0, 2, 1, 3).reshape(B, nblk, -1)                            # (B, n, F*M), zero where invalid
        for b in range(B):
            for q in range(nblk):
                np.add.at(Xh[b, q], flat_idx, contrib[b, q])
        x = np.fft.irfft(Xh, n=g['N'], axis=-1)
        return x.reshape(B, 1, nblk * g['N'])


class WrappedNSGT:
    """
    The transform for conventions whose windows may LEAVE the open positive half-spectrum (the search hypotheses of
    tools/pin_cqt.py: crop start c_k - M, bandwidth from bin k + 1): everything on the FULL length-N spectrum with indices taken
    modulo N, exactly what a dense torch implementation indexing ``fft(x)[..., range_indices]`` with negative indices does.
    No dense (F x N) array: windows are evaluated on the (F x M) crop grid from their closed form.  For conventions that stay
    inside the positive half this equals DenseNSGT.encode (tests/test_nsgt_dense.py); decode returns the REAL PART of the
    inverse FFT of the scattered one-sided spectrum (x / 2 for such conventions; the wrapper's inf-norm removes any scale).
    ``diagonal='mirrored'`` adds the mirror image of every window (at -j) to the frame-operator diagonal.
    """

    def __init__(self, n_octaves=9, bins_per_octave=60, sample_rate=22050, block_length=66150, conventions=None):
        g = self.geo = geometry(n_octaves, bins_per_octave, sample_rate, int(block_length), conventions)
        cv, N, M, L = g['cv'], g['N'], g['M'], g['L']
        m = np.arange(M, dtype=np.int64)[None, :]
        self.idx = (g['s'][:, None] + m) % N                                              # (F, M) wrapped spectral indices
        n = g['s'][:, None] + m - g['a'][:, None]                                         # position inside the window
        inside = (n >= 0) & (n < L[:, None])
        Lf = L[:, None].astype(np.float64)
        denom = Lf if cv['window'] == 'hann_periodic' else np.maximum(Lf - 1.0, 1.0)
        Wg = np.where(inside, 0.5 - 0.5 * np.cos(2.0 * np.pi * n / denom), 0.0)
        self.Wg = np.where(inside & (L[:, None] == 1), 1.0, Wg)
        D = np.zeros(N)
        np.add.at(D, self.idx.reshape(-1), (self.Wg ** 2).reshape(-1))
        if cv['diagonal'] == 'mirrored':
            np.add.at(D, ((-self.idx) % N).reshape(-1), (self.Wg ** 2).reshape(-1))
        self.D = D
        kept = D > (cv['frame_floor'] if cv['dual'] == 'floored' else 0.0)
        den = D + cv['dual_eps'] if cv['dual'] == 'additive' else np.where(kept, D, 1.0)
        self.Wdg = np.where(kept[self.idx], self.Wg / den[self.idx], 0.0)

    def encode(self, audio):
        g = self.geo
        x = np.asarray(audio, dtype=np.float64)
        B = x.shape[0]
        nblk = x.shape[-1] // g['N']
        X = np.fft.fft(x.reshape(B, nblk, g['N']), axis=-1)
        c = np.fft.ifft(X[:, :, self.idx] * self.Wg, axis=-1)
        return c.transpose(0, 2, 1, 3).reshape(B, 1, g['F'], nblk * g['M'])

    def decode(self, coefficients):
        g = self.geo
        c = np.asarray(coefficients)
        B = c.shape[0]
        nblk = c.shape[-1] // g['M']
        V = np.fft.fft(c.reshape(B, g['F'], nblk, g['M']), axis=-1) * self.Wdg[None, :, None, :]
        Xh = np.zeros((B, nblk, g['N']), dtype=np.complex128)
        flat = self.idx.reshape(-1)
        contrib = V.transpose(0, 2, 1, 3).reshape(B, nblk, -1)
        for b in range(B):
            for q in range(nblk):
                np.add.at(Xh[b, q], flat, contrib[b, q])
        return np.fft.ifft(Xh, axis=-1).real.reshape(B, 1, nblk * g['N'])
