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

import numpy as np

N_FAST = 66150          # block length the HIP FFT is specialised for  (N/2 = 675 * 49)
M_FAST = 1024
FRAME_FLOOR = 1e-3   # minimum frame-operator diagonal for a spectral index to be synthesised


def bin_geometry(n_octaves, bins_per_octave, sample_rate, block_length, power_of_2_length=True):
    n_bins = n_octaves * bins_per_octave
    N = int(block_length)
    f_min = (sample_rate / 2) / (2 ** n_octaves)
    freqs = f_min * 2.0 ** (np.arange(n_bins, dtype=np.float64) / bins_per_octave)
    bandwidths = freqs * (2.0 ** (1.0 / bins_per_octave) - 2.0 ** (-1.0 / bins_per_octave))
    lengths = np.maximum(np.round(bandwidths * N / sample_rate), 1).astype(np.int64)
    M = int(lengths.max())
    if power_of_2_length:
        M = 2 ** int(math.ceil(math.log2(M)))
    positions = np.round(freqs * N / sample_rate).astype(np.int64)
    pad = np.floor(M / 2 - lengths / 2).astype(np.int64)
    start = positions - M // 2
    return dict(n_bins=n_bins, N=N, M=M, freqs=freqs, lengths=lengths, positions=positions, pad=pad, start=start)


def build_plan(n_octaves, bins_per_octave, sample_rate, block_length, power_of_2_length=True):
    """numpy tables (host); see CQT._device_plan for the device copies."""
    g = bin_geometry(n_octaves, bins_per_octave, sample_rate, block_length, power_of_2_length)
    N, M, F = g['N'], g['M'], g['n_bins']
    lengths = g['lengths']
    win_off = np.zeros(F + 1, dtype=np.int64)
    win_off[1:] = np.cumsum(lengths)
    total = int(win_off[-1])
    window = np.zeros(total)
    spec_index = np.zeros(total, dtype=np.int64)
    for k in range(F):
        L = int(lengths[k])
        n = np.arange(L, dtype=np.float64)
        window[win_off[k]:win_off[k + 1]] = (0.5 - 0.5 * np.cos(2 * np.pi * n / L)) if L > 1 else 1.0
        spec_index[win_off[k]:win_off[k + 1]] = g['start'][k] + g['pad'][k] + np.arange(L)
    if spec_index.min() <= 0 or spec_index.max() >= N // 2:
        raise ValueError('NSGT window leaves the open positive half-spectrum')
    diag = np.zeros(N // 2 + 1)
    np.add.at(diag, spec_index, window ** 2)
    # indices whose total window energy is below FRAME_FLOOR are not synthesised (band edges: the canonical
    # dual 1/w would reach ~6e4 there and amplify coefficient noise into audible sinusoids)
    covered = diag > FRAME_FLOOR
    dual = np.where(covered[spec_index], window / np.where(covered, diag, 1.0)[spec_index], 0.0)

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
    plan.update(sum_len=total, win_off=win_off, window=window, dual=dual, spec_index=spec_index,
                bin_tab=bin_tab.astype(np.int32), gat_off=gat_off.astype(np.int32),
                gat_idx=order.astype(np.int32), covered=covered)
    if N == N_FAST and M == M_FAST:
        # four-step twiddles W_Nc^(n2 k1) laid out [n2][k1] (49 x 675) so the row kernel reads them coalesced
        n2k1 = np.outer(np.arange(49, dtype=np.float64), np.arange(675, dtype=np.float64)).reshape(-1)
        ang = -2.0 * np.pi * n2k1 / (N // 2)
        plan.update(tw675=tw(675), tw49=tw(49), twNc=np.stack([np.cos(ang), np.sin(ang)], axis=1),
                    twN=tw(N, N // 2 + 1), tw1024=tw(1024))
    return plan
