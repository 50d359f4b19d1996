"""
Scoring next to the hot path: what reference ``experiments/evaluate.py:113-127`` computes
per track from the outputs of the path -- frame-level multi-pitch precision / recall / F1 and the reconstruction SDR.

The reference delegates both to third-party packages that are absent from this image:
  * ``mir_eval.multipitch.evaluate(ref_time, ref_freqs, est_time, est_freqs, window=0.5)`` called from
    ``timbre_trap/utils/experiments.py:376-378``;
  * ``torchmetrics.audio.SignalDistortionRatio()`` (defaults) called from ``experiments/evaluate.py:40,123``.
``multipitch_metrics`` and ``signal_distortion_ratio`` restate their published algorithms (mir_eval's multipitch module after
Poliner & Ellis 2007 / Bay et al. 2009; the filtered SDR of Scheibler 2021, "SDR -- medium rare with fast computations", as
implemented by torchmetrics / fast_bss_eval).  PARITY UNPINNED against the packages themselves (neither is installed and there
is no network); pinned instead by known-answer tests (tests/test_metrics.py) and, for the SDR, by an independent dense
least-squares derivation of the same quantity.  Host code in float64, as in the reference (NumPy lists / CPU tensors).
"""

import sys
from copy import deepcopy

import numpy as np

__all__ = ['resample_multipitch', 'frequencies_to_midi', 'match_count', 'multipitch_metrics', 'MultipitchEvaluator',
           'signal_distortion_ratio']

MAX_FREQ, MIN_FREQ = 5000.0, 20.0            # mir_eval.multipitch validation limits (evaluate.py:44-48 masks bins above 5 kHz)


def resample_multipitch(times, frequencies, target_times):
    """
    Frame lists re-read at ``target_times`` by nearest neighbour (mir_eval.multipitch.resample_multipitch): ties go to the
    earlier frame; targets outside [times[0], times[-1]] get an empty frame.
    """
    target_times = np.asarray(target_times, dtype=np.float64)
    if target_times.size == 0:
        return []
    times = np.asarray(times, dtype=np.float64)
    if times.size == 0:
        return [np.array([])] * len(target_times)
    n = len(frequencies)
    half = times / 2.0
    mids = half[1:] + half[:-1]
    idx = np.clip(np.searchsorted(mids, target_times, side='left'), 0, n - 1)
    idx = np.where((target_times < times[0]) | (target_times > times[-1]), n, idx)
    vals = list(frequencies) + [np.array([])]
    return [vals[int(i)] for i in idx]


def frequencies_to_midi(frequencies, ref_frequency=440.0):
    """Hz -> (fractional) MIDI note numbers, frame by frame."""
    return [69.0 + 12.0 * np.log2(np.asarray(f, dtype=np.float64) / ref_frequency) for f in frequencies]


def midi_to_chroma(frequencies_midi):
    """Wrap MIDI numbers into one octave (mod 12)."""
    return [np.mod(f, 12) for f in frequencies_midi]


def _max_matching(ref, est, window, chroma):
    """
    Size of a maximum matching between reference and estimated pitches of one frame, a pair being admissible when the
    pitches are at most ``window`` semitones apart (``mir_eval.util.match_events``: |ref - est| <= window, with the
    octave-wrapped distance min(d, 12 - d) in chroma mode).  Augmenting paths (Kuhn): frames hold a handful of pitches.
    """
    if len(ref) == 0 or len(est) == 0:
        return 0
    diff = np.abs(np.subtract.outer(np.asarray(ref, dtype=np.float64), np.asarray(est, dtype=np.float64)))
    if chroma:
        diff = np.mod(diff, 12)
        diff = np.minimum(diff, 12 - diff)
    adj = diff <= window
    match_est = -np.ones(len(est), dtype=np.int64)

    def augment(r, seen):
        for e in np.flatnonzero(adj[r]):
            if seen[e]:
                continue
            seen[e] = True
            if match_est[e] < 0 or augment(match_est[e], seen):
                match_est[e] = r
                return True
        return False
    count = 0
    for r in range(len(ref)):
        if augment(r, np.zeros(len(est), dtype=bool)):
            count += 1
    return count


def match_count(ref_midi, est_midi, window=0.5, chroma=False):
    """True positives per frame."""
    return np.array([_max_matching(r, e, window, chroma) for r, e in zip(ref_midi, est_midi)], dtype=np.float64)


def _prf_accuracy(tp, n_ref, n_est):
    tps, nr, ne = tp.sum(), n_ref.sum(), n_est.sum()
    precision = tps / ne if ne > 0 else 0.0
    recall = tps / nr if nr > 0 else 0.0
    denom = (n_est + n_ref - tp).sum()
    accuracy = tps / denom if denom > 0 else 0.0
    return float(precision), float(recall), float(accuracy)


def _error_scores(tp, n_ref, n_est):
    nr = n_ref.sum()
    if nr == 0:
        return 0.0, 0.0, 0.0, 0.0
    e_sub = (np.minimum(n_ref, n_est) - tp).sum() / nr
    e_miss = np.maximum(n_ref - n_est, 0).sum() / nr
    e_fa = np.maximum(n_est - n_ref, 0).sum() / nr
    e_tot = (np.maximum(n_ref, n_est) - tp).sum() / nr
    return float(e_sub), float(e_miss), float(e_fa), float(e_tot)


def multipitch_metrics(ref_time, ref_freqs, est_time, est_freqs, window=0.5):
    """
    The fourteen scores of ``mir_eval.multipitch.evaluate`` (same key names): estimates are re-read on the REFERENCE time base
    by nearest neighbour, every frame's pitches go to MIDI, and true positives per frame are the size of a maximum matching of
    reference and estimated pitches at most ``window`` semitones apart (octave-wrapped for the Chroma scores).
    """
    ref_time = np.asarray(ref_time, dtype=np.float64)
    est_time = np.asarray(est_time, dtype=np.float64)
    if len(ref_time) != len(ref_freqs) or len(est_time) != len(est_freqs):
        raise ValueError('time and frequency lists must have the same number of frames')
    for name, freqs in (('reference', ref_freqs), ('estimate', est_freqs)):
        for f in freqs:
            f = np.asarray(f)
            if f.size and (f.max() > MAX_FREQ or f.min() < MIN_FREQ):
                raise ValueError('%s frequencies must lie in [%g, %g] Hz' % (name, MIN_FREQ, MAX_FREQ))
    keys = ['Precision', 'Recall', 'Accuracy', 'Substitution Error', 'Miss Error', 'False Alarm Error', 'Total Error']
    out = {k: 0.0 for k in keys + ['Chroma ' + k for k in keys]}
    if len(ref_time) == 0 or len(est_time) == 0:
        return out
    est_resampled = resample_multipitch(est_time, est_freqs, ref_time)
    ref_midi, est_midi = frequencies_to_midi(ref_freqs), frequencies_to_midi(est_resampled)
    n_ref = np.array([len(f) for f in ref_midi], dtype=np.float64)
    n_est = np.array([len(f) for f in est_midi], dtype=np.float64)
    for prefix, chroma in (('', False), ('Chroma ', True)):
        r = midi_to_chroma(ref_midi) if chroma else ref_midi
        e = midi_to_chroma(est_midi) if chroma else est_midi
        tp = match_count(r, e, window, chroma)
        p, rc, a = _prf_accuracy(tp, n_ref, n_est)
        es, em, ef, et = _error_scores(tp, n_ref, n_est)
        for k, v in zip(keys, (p, rc, a, es, em, ef, et)):
            out[prefix + k] = v
    return out


class MultipitchEvaluator(object):
    """
    Result tracker with the reference's interface (``timbre_trap/utils/experiments.py:277-396``): ``evaluate`` returns the
    scores above under lower-case keys prefixed ``mpe/`` plus ``mpe/f1-score = 2 P R / (P + R + eps)``; ``append_results`` /
    ``average_results`` accumulate per-track dictionaries and give rounded means and standard deviations.
    """

    def __init__(self, tolerance=0.5):
        self.tolerance = tolerance
        self.results = None
        self.reset_results()

    def reset_results(self):
        self.results = {}

    def append_results(self, results):
        for key in results.keys():
            if key in self.results.keys():
                self.results[key] = np.append(self.results[key], results[key])
            else:
                self.results[key] = np.array([results[key]])

    def average_results(self):
        mean, std_dev = deepcopy(self.results), deepcopy(self.results)
        for key in self.results.keys():
            mean[key] = round(np.mean(mean[key]), 5)
            std_dev[key] = round(np.std(std_dev[key]), 5)
        return mean, std_dev

    def evaluate(self, times_est, multi_pitch_est, times_ref, multi_pitch_ref):
        scores = multipitch_metrics(times_ref, multi_pitch_ref, times_est, multi_pitch_est, window=self.tolerance)
        results = {k.lower(): v for k, v in scores.items()}
        pr, rc = results['precision'], results['recall']
        results['f1-score'] = 2 * pr * rc / (pr + rc + sys.float_info.epsilon)
        return {'mpe/' + k: v for k, v in results.items()}


# ---- SDR --------------------------------------------------------------------------------------------------------------

def _next_pow2(n):
    return 1 << int(np.ceil(np.log2(max(int(n), 1))))


def signal_distortion_ratio(preds, target, filter_length=512, zero_mean=False, load_diag=None):
    """
    Signal-to-distortion ratio in dB, the distortion being what is left of ``preds`` after projection onto the span of
    ``filter_length`` delayed copies of ``target`` (BSS-eval's SDR, computed the fast way):

        both signals scaled to unit norm;  r = autocorrelation of target (lags 0 .. L-1),  b = cross-correlation target -> preds;
        solve  Toeplitz(r) h = b ;   coherence = b . h ;   SDR = 10 log10(coherence / (1 - coherence)).

    ``preds`` / ``target``: arrays or tensors (..., time); float64 throughout; returns an ndarray of the leading shape (a float
    for 1-D input).  Mirrors ``torchmetrics.functional.audio.signal_distortion_ratio`` with its defaults (dense solve).
    """
    p = np.asarray(preds.detach().cpu() if hasattr(preds, 'detach') else preds, dtype=np.float64)
    t = np.asarray(target.detach().cpu() if hasattr(target, 'detach') else target, dtype=np.float64)
    if p.shape != t.shape:
        raise ValueError('preds and target must have the same shape')
    if zero_mean:
        p = p - p.mean(axis=-1, keepdims=True)
        t = t - t.mean(axis=-1, keepdims=True)
    t = t / np.maximum(np.linalg.norm(t, axis=-1, keepdims=True), 1e-6)
    p = p / np.maximum(np.linalg.norm(p, axis=-1, keepdims=True), 1e-6)
    n_fft = _next_pow2(p.shape[-1] + t.shape[-1] - 1)
    tf = np.fft.rfft(t, n=n_fft, axis=-1)
    r0 = np.fft.irfft(tf.real ** 2 + tf.imag ** 2, n=n_fft, axis=-1)[..., :filter_length]
    b = np.fft.irfft(np.conj(tf) * np.fft.rfft(p, n=n_fft, axis=-1), n=n_fft, axis=-1)[..., :filter_length]
    if load_diag is not None:
        r0 = r0.copy()
        r0[..., 0] += load_diag
    lead = r0.shape[:-1]
    r0f, bf = r0.reshape(-1, filter_length), b.reshape(-1, filter_length)
    idx = np.abs(np.subtract.outer(np.arange(filter_length), np.arange(filter_length)))
    out = np.empty(r0f.shape[0])
    for i in range(r0f.shape[0]):
        sol = np.linalg.solve(r0f[i][idx], bf[i])
        coh = float(np.dot(bf[i], sol))
        ratio = coh / (1.0 - coh)
        out[i] = 10.0 * np.log10(ratio) if ratio > 0 else -np.inf
    return float(out[0]) if not lead else out.reshape(lead)
