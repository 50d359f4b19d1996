"""
Target generation and its inverse on the device: the two static methods of reference
``timbre_trap/datasets/PitchDataset.py`` that sit directly before (``multi_pitch_to_activations``, :233-307) and after
(``activations_to_multi_pitch``, :309-348) the hot path.

The ragged Python lists are flattened on the host (nearest-bin lookup included: a ``searchsorted`` over ~10^3 values);
the F x T arithmetic -- scatter, Gaussian blur along frequency, renormalisation, clip; peak picking and thresholding --
runs in HIP kernels (tt_target_activations in float64, bit-identical to the reference's SciPy calls; tt_peak_pick).
"""

import warnings

import numpy as np
import torch

from .. import _hip

__all__ = ['multi_pitch_to_activations', 'activations_to_multi_pitch', 'hz_to_midi', 'midi_to_hz']


def hz_to_midi(frequencies):
    """librosa.hz_to_midi: 12 (log2 f - log2 440) + 69."""
    return 12 * (np.log2(np.asanyarray(frequencies, dtype=np.float64)) - np.log2(440.0)) + 69


def midi_to_hz(notes):
    """librosa.midi_to_hz: 440 * 2^((m - 69) / 12)."""
    return 440.0 * (2.0 ** ((np.asanyarray(notes, dtype=np.float64) - 69.0) / 12.0))


def _gaussian_weights(sigma, truncate=4.0):
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return phi / phi.sum(), radius


def multi_pitch_to_activations(multi_pitch, midi_freqs, n_bins_blur_decay=2.5, device='cuda', return_tensor=False):
    """
    List of per-frame pitch arrays (Hz) -> (F, T) float64 activations: 1 at the nearest bin of every in-range pitch
    (ties to the lower bin, like ``interp1d(kind='nearest')``), optionally blurred along frequency with a Gaussian of
    sigma = 2 n_bins_blur_decay / 5 bins, renormalised so every annotated bin is >= 1, and clipped to [0, 1].
    Returns an ndarray like the reference (``return_tensor=True``: the float64 device tensor, no download).
    """
    midi_freqs = np.asarray(midi_freqs, dtype=np.float64)
    F, T = len(midi_freqs), len(multi_pitch)
    lb, ub = np.min(midi_freqs), np.max(midi_freqs)
    mids = (midi_freqs[1:] + midi_freqs[:-1]) / 2.0
    bins, frames, num_nonzero = [], [], 0
    for t, p in enumerate(multi_pitch):
        p = np.asarray(p, dtype=np.float64)
        m = hz_to_midi(p[p != 0])
        num_nonzero += len(m)
        m = m[np.logical_and(m >= lb, m <= ub)]
        if len(m):
            bins.append(np.searchsorted(mids, m, side='left'))
            frames.append(np.full(len(m), t, dtype=np.int64))
    n = int(sum(len(b) for b in bins))
    if n != num_nonzero:
        warnings.warn('Could not fully represent ground-truth with available frequency bins.', RuntimeWarning)
    dev = torch.device(device)
    if dev.type != 'cuda':
        raise RuntimeError('timbre_trap HIP path needs a GPU device (got %s); there is no CPU fallback' % dev)
    out = torch.empty((F, T), dtype=torch.float64, device=dev)
    if F * T == 0:
        return out if return_tensor else out.cpu().numpy()
    radius, w_t, work = 0, None, None
    if n and n_bins_blur_decay:
        w, radius = _gaussian_weights((2 * n_bins_blur_decay) / 5)
        w_t = torch.from_numpy(w).to(dev)
        work = torch.empty((F, T), dtype=torch.float64, device=dev)
    b_t = torch.from_numpy(np.concatenate(bins).astype(np.int32)).to(dev) if n else None
    f_t = torch.from_numpy(np.concatenate(frames).astype(np.int32)).to(dev) if n else None
    with torch.cuda.device(dev):
        _hip.check(_hip.lib().tt_target_activations(_hip.ptr(b_t), _hip.ptr(f_t), n, _hip.ptr(w_t), radius, F, T, _hip.ptr(work),
                                                    _hip.ptr(out), _hip.stream_ptr()), 'tt_target_activations')
    return out if return_tensor else out.cpu().numpy()


def activations_to_multi_pitch(activations, midi_freqs, peaks_only=False, t=0.5):
    """
    (F, T) activations (ndarray or device tensor) -> list of T arrays with the active pitches in Hz: optional strict
    local-peak picking along frequency, then ``>= t``, both in one device pass; the ragged result is assembled on the host.
    """
    from .processing import _device_pick
    mask = _device_pick(activations, t, 2 if peaks_only else 1)
    mask = mask if isinstance(mask, np.ndarray) else mask.cpu().numpy()
    midi_freqs = np.asarray(midi_freqs, dtype=np.float64)
    multi_pitch = [np.empty(0)] * mask.shape[-1]
    for i in np.where(np.sum(mask, axis=-2) > 0)[-1]:
        multi_pitch[i] = midi_to_hz(midi_freqs[np.where(mask[..., i])[-1]])
    return multi_pitch
