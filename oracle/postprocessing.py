"""
ORACLE (test infrastructure only -- never imported by the product path).

Restatement of the host-side helpers next to the hot path:
  - reference timbre_trap/utils/experiments.py:81-141 (CosineWarmup closed form), :144-256 (gradient statistics)
  - reference timbre_trap/utils/processing.py:66-124 (filter_non_peaks, threshold)
Pinned by tests/golden/utils.npz (generated from the imported reference).
"""

import math

import numpy as np


def cosine_warmup_scale(step_index, n_steps):
    """experiments.py:129-141 with last_epoch = step_index: 1 - 0.5 (1 + cos((1 + min(i, N)) pi / (N + 1)))."""
    n_steps = max(0, n_steps)
    curr = 1 + min(step_index, n_steps)
    return 1 - 0.5 * (1 + math.cos(curr * math.pi / (n_steps + 1)))


def gradient_statistics(grads):
    """[sum of L2 norms, mean of L2 norms, max |g|, max L2 norm] over a list of arrays (experiments.py:144-256)."""
    norms = [float(np.sqrt((np.asarray(g, dtype=np.float64) ** 2).sum())) for g in grads]
    return [sum(norms), sum(norms) / len(norms), max(float(np.abs(g).max()) for g in grads), max(norms)]


def filter_non_peaks(arr):
    """processing.py:66-100: keep strict local maxima along axis -2 (zero rows padded at both ends), zero elsewhere."""
    a = np.asarray(arr, dtype=np.float64)
    z = np.zeros(a.shape[:-2] + (1, a.shape[-1]))
    p = np.concatenate((z, a, z), axis=-2)
    mid, up, down = p[..., 1:-1, :], p[..., :-2, :], p[..., 2:, :]
    return np.where((mid > up) & (mid > down), mid, 0.0)


def threshold(arr, t=0.5):
    """processing.py:103-124."""
    return (np.asarray(arr) >= t).astype(np.float64)
