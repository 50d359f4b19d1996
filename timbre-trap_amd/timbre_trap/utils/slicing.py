"""
Excerpt slicing and annotation resampling directly in front of the hot path: the index
arithmetic of reference ``timbre_trap/datasets/AudioDataset.py:88-143`` (``slice_audio``) and
``timbre_trap/datasets/PitchDataset.py:79-137`` (``slice_times``) / ``:194-231`` (``resample_multi_pitch``).

These run in DataLoader worker processes in the reference, so they stay host code (NumPy, no HIP runtime): what they
define is WHICH samples and frames reach the device, bit for bit -- the same ``RandomState`` draws in the same order, the
same float64 offsets, the same nearest-neighbour rule (``scipy.interpolate.interp1d(kind='nearest')``: ties go to the earlier
annotation, targets outside the annotated span take the first / last annotation selected by ``resample_idcs``).
The functions take the state the reference keeps on ``self`` (``cqt``, ``n_secs``, ``rng``, ``sample_rate``) as arguments;
``ExcerptSlicer`` bundles them with the reference's method names.
"""

import numpy as np
import torch

__all__ = ['slice_audio', 'slice_times', 'nearest_indices', 'resample_multi_pitch', 'ExcerptSlicer']


def slice_audio(audio, n_samples=None, offset_s=None, *, n_secs=None, sample_rate=None, rng=None):
    """
    Cut (or zero-pad) ``audio`` (1 x N tensor) to ``n_samples`` samples (reference AudioDataset.py:88-143).

    Returns (audio (1 x n_samples), offset_t): the offset in seconds of the excerpt inside the track; negative when the
    track was shorter than the excerpt and ``-offset_t`` seconds of silence were put in front of it.
    """
    if n_samples is None:
        n_samples = int(n_secs * sample_rate)
    n = audio.size(-1)
    if n >= n_samples:
        start = rng.randint(0, n - n_samples + 1) if offset_s is None else offset_s
        offset_t = start / sample_rate
        audio = audio[..., start: start + n_samples]
    else:
        pad_total = n_samples - n
        pad_left = rng.randint(0, pad_total) if offset_s is None else abs(offset_s)
        offset_t = -pad_left / sample_rate
        audio = torch.nn.functional.pad(audio, (pad_left, pad_total - pad_left))
    return audio, offset_t


def slice_times(times, n_frames=None, offset_t=None, *, cqt, n_secs=None, rng=None, sample_rate=None):
    """
    Cut (or pad with -inf / +inf) frame times to ``n_frames`` frames (reference PitchDataset.py:79-137).

    Returns (times (n_frames), offset_n): offset in frames of the excerpt; with ``offset_t`` given the times are REBUILT on the
    transform's frame grid shifted by ``offset_t`` (not sliced), exactly like the reference.
    """
    if n_frames is None:
        n_frames = cqt.get_expected_frames(cqt.get_expected_samples(n_secs))
    if len(times) >= n_frames:
        if offset_t is None:
            start = rng.randint(0, times.size - n_frames + 1)
            offset_n = start
            times = times[start: start + n_frames]
        else:
            times = cqt.get_times(n_frames) + offset_t
            offset_n = offset_t * (cqt.sample_rate / cqt.hop_length)
    else:
        pad_total = n_frames - len(times)
        if offset_t is None:
            pad_left = rng.randint(0, pad_total)
        else:
            pad_left = round(abs(offset_t) * sample_rate / cqt.hop_length)
        offset_n = -pad_left
        times = np.pad(times, (pad_left, 0), constant_values=-np.inf)
        times = np.pad(times, (0, pad_total - pad_left), constant_values=np.inf)
    return times, offset_n


def nearest_indices(source_times, target_times, below, above):
    """
    Index of the source time nearest to every target time: the rule of ``scipy.interpolate.interp1d(kind='nearest',
    assume_sorted=True, bounds_error=False, fill_value=(below, above))`` -- interval midpoints decide, an exact midpoint goes to
    the EARLIER source, targets before the first / after the last source time (including -inf / +inf padding) take ``below`` /
    ``above``.
    """
    src = np.asarray(source_times, dtype=np.float64)
    tgt = np.asarray(target_times, dtype=np.float64)
    half = src / 2.0
    mids = half[1:] + half[:-1]                       # the same floating-point expression SciPy evaluates
    idx = np.searchsorted(mids, tgt, side='left').astype(np.int64)
    idx = np.clip(idx, 0, len(src) - 1)
    idx = np.where(tgt < src[0], below, idx)
    idx = np.where(tgt > src[-1], above, idx)
    return idx


def resample_multi_pitch(_times, _multi_pitch, times, resample_idcs=(0, -1)):
    """
    Frame-level pitch lists re-read on a new time grid by nearest neighbour (reference PitchDataset.py:194-231).
    ``resample_idcs`` selects which original frames stand in for targets before / after the annotated span.
    """
    n = len(_times)
    original = np.arange(n)
    below, above = int(original[resample_idcs[0]]), int(original[resample_idcs[-1]])
    idx = nearest_indices(_times, times, below, above)
    return [_multi_pitch[int(i)] for i in idx]


class ExcerptSlicer:
    """
    The slicing state of one reference dataset object (``cqt``, ``n_secs``, ``sample_rate``, ``rng = RandomState(seed)``,
    ``resample_idcs``) with the reference's method names, for callers that feed the HIP path without the reference's loaders.
    """

    def __init__(self, cqt, n_secs=None, sample_rate=None, seed=0, resample_idcs=None):
        self.cqt = cqt
        self.n_secs = n_secs
        self.sample_rate = cqt.sample_rate if sample_rate is None else sample_rate
        self.rng = np.random.RandomState(seed)
        self.resample_idcs = [0, -1] if resample_idcs is None else resample_idcs

    def slice_audio(self, audio, n_samples=None, offset_s=None):
        return slice_audio(audio, n_samples, offset_s, n_secs=self.n_secs, sample_rate=self.sample_rate, rng=self.rng)

    def slice_times(self, times, n_frames=None, offset_t=None):
        return slice_times(times, n_frames, offset_t, cqt=self.cqt, n_secs=self.n_secs, rng=self.rng, sample_rate=self.sample_rate)

    def resample_multi_pitch(self, _times, _multi_pitch, times):
        return resample_multi_pitch(_times, _multi_pitch, times, self.resample_idcs)
