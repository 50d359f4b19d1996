"""
ORACLE (test infrastructure only -- never imported by the product path).

Restatement of the target generation on either side of the hot path:
reference timbre_trap/datasets/PitchDataset.py:233-307 (multi_pitch_to_activations) and :309-348
(activations_to_multi_pitch), with the two SciPy routines they call written out:
  scipy.interpolate.interp1d(kind='nearest')  ->  searchsorted over the midpoints of the grid, ties to the LOWER bin
  scipy.ndimage.gaussian_filter1d(mode='constant')  ->  radius int(4 sigma + 0.5), weights exp(-x^2 / (2 sigma^2)) normalised,
                                                      symmetric correlation  c w0 + sum_j (x[-j] + x[+j]) w_j  in that order
Pinned by tests/golden/targets.npz (generated from the imported reference).
"""

import numpy as np


def hz_to_midi(f):
    return 12 * (np.log2(np.asanyarray(f, dtype=np.float64)) - np.log2(440.0)) + 69


def midi_to_hz(m):
    return 440.0 * (2.0 ** ((np.asanyarray(m, dtype=np.float64) - 69.0) / 12.0))


def nearest_bins(midi, midi_freqs):
    mids = (midi_freqs[1:] + midi_freqs[:-1]) / 2.0
    return np.searchsorted(mids, midi, side='left').astype(np.int64)


def gaussian_weights(sigma, truncate=4.0):
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return phi / phi.sum(), radius


def blur_rows(a, w, r):
    """Correlation along axis 0 with zero padding, in SciPy's symmetric-kernel order."""
    F = a.shape[0]
    p = np.zeros((F + 2 * r,) + a.shape[1:])
    p[r:r + F] = a
    out = p[r:r + F] * w[r]
    for j in range(-r, 0):
        out = out + (p[r + j:r + j + F] + p[r - j:r - j + F]) * w[r + j]
    return out


def multi_pitch_to_activations(multi_pitch, midi_freqs, n_bins_blur_decay=2.5):
    midi_freqs = np.asarray(midi_freqs, dtype=np.float64)
    act = np.zeros((len(midi_freqs), len(multi_pitch)))
    lb, ub = midi_freqs.min(), midi_freqs.max()
    bins, frames = [], []
    for t, p in enumerate(multi_pitch):
        p = np.asarray(p, dtype=np.float64)
        m = hz_to_midi(p[p != 0])
        m = m[(m >= lb) & (m <= ub)]
        bins.append(nearest_bins(m, midi_freqs))
        frames.append(np.full(len(m), t, dtype=np.int64))
    bins, frames = np.concatenate(bins), np.concatenate(frames)
    if len(bins):
        act[bins, frames] = 1
        if n_bins_blur_decay:
            w, r = gaussian_weights((2 * n_bins_blur_decay) / 5)
            act = blur_rows(act, w, r)
            act = act / np.min(act[bins, frames])
            act = np.clip(act, 0.0, 1.0)
    return act


def activations_to_multi_pitch(activations, midi_freqs, peaks_only=False, t=0.5):
    from .postprocessing import filter_non_peaks, threshold
    a = np.asarray(activations)
    if peaks_only:
        a = filter_non_peaks(a)
    a = threshold(a, t)
    midi_freqs = np.asarray(midi_freqs, dtype=np.float64)
    return [midi_to_hz(midi_freqs[np.where(a[..., i])[-1]]) if a[..., i].sum() > 0 else np.empty(0) for i in range(a.shape[-1])]
