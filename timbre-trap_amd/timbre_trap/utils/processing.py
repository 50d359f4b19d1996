"""
Post-processing of activations next to the hot path: reference
``timbre_trap/utils/processing.py:66-124``.  Device tensors go through one HIP kernel (tt_peak_pick) so that
``evaluate()`` can keep the activations on the GPU; NumPy arrays (the reference's own calling convention) are
uploaded and go through the same kernel.
"""

import numpy as np
import torch

from .. import _hip

__all__ = ['to_array', 'debug_nans', 'filter_non_peaks', 'threshold', 'peaks_above']


def to_array(tensor):
    """Tensor -> ndarray on the host (reference processing.py:17-33)."""
    return tensor.cpu().detach().numpy()


def debug_nans(tensor, tag='tensor'):
    """True (and a warning) when the tensor holds a NaN (reference processing.py:36-63)."""
    import warnings
    contains = bool(torch.isnan(tensor).any())
    if contains:
        warnings.warn(f'{tag} contains NaNs!!!')
    return contains


def _device_pick(x, thr, mode, f_valid=0):
    """
    Tensors stay on the device and come back as float32 tensors; ndarrays (the reference's calling convention) are
    uploaded, processed by the same kernel and returned as float64 ndarrays like the reference returns.  The kernel
    compares in float32, which is the precision the activations are produced in.
    """
    as_array = not isinstance(x, torch.Tensor)
    if as_array:
        x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to('cuda')
    _hip.require_cuda(x)
    xc = x.detach().to(torch.float32).contiguous()
    if xc.dim() < 2:
        raise ValueError('expected (..., F, T) activations')
    F, T = xc.size(-2), xc.size(-1)
    out = torch.empty_like(xc)
    if xc.numel():
        _hip.check(_hip.lib().tt_peak_pick(_hip.ptr(xc), _hip.ptr(out), xc.numel() // (F * T), F, T, float(thr), mode,
                                           int(f_valid), _hip.stream_ptr()), 'tt_peak_pick')
    return out.cpu().numpy().astype(np.float64) if as_array else out


def filter_non_peaks(_arr):
    """Keep strict local maxima along the second-to-last axis (zero rows assumed beyond both ends), zero elsewhere."""
    return _device_pick(_arr, 0.0, 0)


def threshold(_arr, t=0.5):
    """1 where the data reaches the threshold, else 0."""
    return _device_pick(_arr, t, 1)


def peaks_above(activations, t=0.5, n_valid_bins=0):
    """
    threshold(filter_non_peaks(x), t) in one device pass (what evaluate() feeds to the multi-pitch conversion).
    ``n_valid_bins`` > 0 first zeroes the rows from that bin upwards -- the ``activations[valid_freqs] = 0`` step of
    reference experiments/evaluate.py:107-112 (bins above mir_eval's 5 kHz limit: 472 of 540 at the default geometry).
    """
    return _device_pick(activations, t, 2, n_valid_bins)
