"""
Host-side plan of the NSGT constant-Q transform: every convention of the transform as tables
(computed once in float64, shipped to the device as float32 / int32) that the HIP kernels in
csrc/cqt.hip consume.  See include/ttrap.h (tt_cqt_plan) for the table contract.

The transform stands in for ``cqt_pytorch.CQT(num_octaves, num_bins_per_octave, sample_rate,
block_length, power_of_2_length=True)`` constructed at reference
timbre_trap/framework/cqtwrapper.py:31-35: a non-stationary Gabor transform with one periodic-Hann
window per geometrically spaced bin applied to the block's spectrum, an M-point inverse FFT per
bin, and the canonical dual windows ("painless" case) for synthesis.
"""

import math
from dataclasses import dataclass, replace

import numpy as np

N_FAST = 66150          # block length the HIP FFT is specialised for  (N/2 = 675 * 49)
M_FAST = 1024
FRAME_FLOOR = 1e-3   # minimum frame-operator diagonal for a spectral index to be synthesised (default dual = 'floored')


@dataclass(frozen=True)
class NSGTConventions:
    """
    Every free choice of the transform that the reference leaves to ``cqt_pytorch`` (absent here, so UNPINNED -- see
    DESIGN.md section 2).  All of them are table-level: switching one changes the plan, never a kernel.  The defaults are the
    recalled ``cqt_pytorch`` behaviour plus ONE deliberate deviation (``dual='floored'``, documented in INTEGRATION.md);
    ``tools/pin_cqt.py`` searches this space for the combination that reproduces ``cqt_pytorch`` when that package is available.

      window           'hann_periodic' (torch.hann_window default: 0.5 - 0.5 cos(2 pi n / L)) | 'hann_symmetric' (... / (L - 1))
      length_rounding  how  Omega_k N / sr  becomes the window length L_k:   'round' (half to even) | 'floor' | 'ceil'
      centre_rounding  how  f_k N / sr      becomes the centre index c_k:    'round' (half to even) | 'floor' | 'ceil'
      crop_alignment   where the M-sample crop sits: 'centred' (crop starts at c_k - M/2, window at floor(M/2 - L_k/2) inside it)
                       | 'window_start' (crop starts at the window's first sample, c_k - floor(L_k/2); a per-bin linear phase)
      dual             'floored'   canonical dual g_k / D where the frame-operator diagonal D > frame_floor, 0 elsewhere
                       | 'canonical'  g_k / D wherever D > 0 (exact inverse on the covered band; amplifies coefficient noise by up
                         to ~6e4 at the few indices only the tail of one window reaches)
                       | 'additive'  g_k / (D + dual_eps) wherever any window reaches -- the regularised inverse some NSGT
                         implementations use instead of a floor (a plausible cqt_pytorch form, unverified)
      min_length       lower bound of L_k (1: the lowest bins keep a one-sample window)

    The frame-operator diagonal D is taken over the positive-frequency windows only: with every window inside the OPEN positive
    half-spectrum (build_plan raises otherwise) the mirrored negative-frequency windows of a real signal never reach an index
    of that half, so a diagonal that includes them is the same table -- that candidate form needs no switch here.
    """
    window: str = 'hann_periodic'
    length_rounding: str = 'round'
    centre_rounding: str = 'round'
    crop_alignment: str = 'centred'
    dual: str = 'floored'
    frame_floor: float = FRAME_FLOOR
    min_length: int = 1
    dual_eps: float = 1e-8

    def replace(self, **kw):
        return replace(self, **kw)


DEFAULT_CONVENTIONS = NSGTConventions()
_ROUND = {'round': np.round, 'floor': np.floor, 'ceil': np.ceil}


def hann(L, family):
    """The analysis window of one bin, L samples."""
    if L <= 1:
        return np.ones(max(L, 1))
    n = np.arange(L, dtype=np.float64)
    if family == 'hann_periodic':
        return 0.5 - 0.5 * np.cos(2 * np.pi * n / L)
    if family == 'hann_symmetric':
        return 0.5 - 0.5 * np.cos(2 * np.pi * n / (L - 1))
    raise ValueError('unknown window family %r' % (family,))


def bin_geometry(n_octaves, bins_per_octave, sample_rate, block_length, power_of_2_length=True, conventions=DEFAULT_CONVENTIONS):
    cv = conventions
    n_bins = n_octaves * bins_per_octave
    N = int(block_length)
    f_min = (sample_rate / 2) / (2 ** n_octaves)
    freqs = f_min * 2.0 ** (np.arange(n_bins, dtype=np.float64) / bins_per_octave)
    bandwidths = freqs * (2.0 ** (1.0 / bins_per_octave) - 2.0 ** (-1.0 / bins_per_octave))
    lengths = np.maximum(_ROUND[cv.length_rounding](bandwidths * N / sample_rate), cv.min_length).astype(np.int64)
    M = int(lengths.max())
    if power_of_2_length:
        M = 2 ** int(math.ceil(math.log2(M)))
    positions = _ROUND[cv.centre_rounding](freqs * N / sample_rate).astype(np.int64)
    if cv.crop_alignment == 'centred':
        pad = np.floor(M / 2 - lengths / 2).astype(np.int64)
        start = positions - M // 2
    elif cv.crop_alignment == 'window_start':
        pad = np.zeros(n_bins, dtype=np.int64)
        start = positions - lengths // 2
    else:
        raise ValueError('unknown crop alignment %r' % (cv.crop_alignment,))
    return dict(n_bins=n_bins, N=N, M=M, freqs=freqs, lengths=lengths, positions=positions, pad=pad, start=start)


def build_plan(n_octaves, bins_per_octave, sample_rate, block_length, power_of_2_length=True, conventions=DEFAULT_CONVENTIONS):
    """numpy tables (host); see CQT._device_plan for the device copies."""
    cv = conventions
    g = bin_geometry(n_octaves, bins_per_octave, sample_rate, block_length, power_of_2_length, cv)
    N, M, F = g['N'], g['M'], g['n_bins']
    lengths = g['lengths']
    win_off = np.zeros(F + 1, dtype=np.int64)
    win_off[1:] = np.cumsum(lengths)
    total = int(win_off[-1])
    window = np.zeros(total)
    spec_index = np.zeros(total, dtype=np.int64)
    for k in range(F):
        L = int(lengths[k])
        window[win_off[k]:win_off[k + 1]] = hann(L, cv.window)
        spec_index[win_off[k]:win_off[k + 1]] = g['start'][k] + g['pad'][k] + np.arange(L)
    if spec_index.min() <= 0 or spec_index.max() >= N // 2:
        raise ValueError('NSGT window leaves the open positive half-spectrum')
    diag = np.zeros(N // 2 + 1)
    np.add.at(diag, spec_index, window ** 2)
    # 'floored' (default): indices whose total window energy is below the floor are not synthesised (band edges: the
    # canonical dual 1/w would reach ~6e4 there and amplify coefficient noise into audible sinusoids); 'canonical': every
    # index any window reaches is inverted exactly
    if cv.dual not in ('floored', 'canonical', 'additive'):
        raise ValueError('unknown dual window rule %r' % (cv.dual,))
    covered = diag > (cv.frame_floor if cv.dual == 'floored' else 0.0)
    denom = diag + cv.dual_eps if cv.dual == 'additive' else np.where(covered, diag, 1.0)
    dual = np.where(covered[spec_index], window / denom[spec_index], 0.0)

    # CSR: spectral index j -> ragged positions that land on it (deterministic overlap-add)
    order = np.argsort(spec_index, kind='stable')
    counts = np.bincount(spec_index, minlength=N // 2 + 1)
    gat_off = np.zeros(N // 2 + 2, dtype=np.int64)
    gat_off[1:] = np.cumsum(counts)

    bin_tab = np.stack([g['start'] + g['pad'], g['pad'], lengths, win_off[:-1]], axis=1)

    def tw(n, count=None):
        j = np.arange(n if count is None else count, dtype=np.float64)
        a = -2.0 * np.pi * j / n
        return np.stack([np.cos(a), np.sin(a)], axis=1)

    plan = dict(g)
    plan.update(conventions=cv, frame_diag=diag, sum_len=total, win_off=win_off, window=window, dual=dual, spec_index=spec_index,
                bin_tab=bin_tab.astype(np.int32), gat_off=gat_off.astype(np.int32),
                gat_idx=order.astype(np.int32), covered=covered)
    if N == N_FAST and M == M_FAST:
        # four-step twiddles W_Nc^(n2 k1) laid out [n2][k1] (49 x 675) so the row kernel reads them coalesced
        n2k1 = np.outer(np.arange(49, dtype=np.float64), np.arange(675, dtype=np.float64)).reshape(-1)
        ang = -2.0 * np.pi * n2k1 / (N // 2)
        plan.update(tw675=tw(675), tw49=tw(49), twNc=np.stack([np.cos(ang), np.sin(ang)], axis=1),
                    twN=tw(N, N // 2 + 1), tw1024=tw(1024))
    return plan
