"""
f2 row: the scoring restatements (timbre_trap/utils/metrics.py).  mir_eval and torchmetrics are absent from the image, so the
checks are known-answer tests worked by hand from the published definitions, an independent brute-force matcher, and -- for
the SDR -- an independent dense least-squares derivation of the same quantity (projection onto 512 delayed copies).
"""

import itertools
import sys

import numpy as np
import pytest


def _hz(m):
    return 440.0 * 2.0 ** ((np.asarray(m, dtype=np.float64) - 69.0) / 12.0)


def test_multipitch_scores_known_answers():
    from timbre_trap.utils import multipitch_metrics
    t = np.arange(4) * 0.01
    ref = [_hz([60, 64, 67]), _hz([60]), np.array([]), _hz([72, 76])]
    est = [_hz([60.2, 64.6, 55]), _hz([60.5, 61]), _hz([50]), np.array([])]
    # frame 0: 60.2 matches 60 (0.2 <= 0.5), 64.6 does not match 64 (0.6), 55 nothing          -> TP 1, n_ref 3, n_est 3
    # frame 1: 60.5 matches 60 (exactly on the inclusive boundary), 61 does not                  -> TP 1, n_ref 1, n_est 2
    # frame 2: no reference                                                                      -> TP 0, n_ref 0, n_est 1
    # frame 3: no estimate                                                                       -> TP 0, n_ref 2, n_est 0
    s = multipitch_metrics(t, ref, t, est, window=0.5)
    assert s['Precision'] == pytest.approx(2 / 6) and s['Recall'] == pytest.approx(2 / 6)
    assert s['Accuracy'] == pytest.approx(2 / (6 + 6 - 2))
    assert s['Substitution Error'] == pytest.approx(((3 - 1) + (1 - 1) + 0 + 0) / 6)
    assert s['Miss Error'] == pytest.approx((0 + 0 + 0 + 2) / 6)
    assert s['False Alarm Error'] == pytest.approx((0 + 1 + 1 + 0) / 6)
    assert s['Total Error'] == pytest.approx(((3 - 1) + (2 - 1) + 1 + 2) / 6)
    assert s['Total Error'] == pytest.approx(s['Substitution Error'] + s['Miss Error'] + s['False Alarm Error'])
    # chroma: 55 (= 7 mod 12) now matches 67 (= 7 mod 12); 61 vs 60: 1 semitone, no
    assert s['Chroma Precision'] == pytest.approx(3 / 6) and s['Chroma Recall'] == pytest.approx(3 / 6)
    assert set(s) == {p + k for p in ('', 'Chroma ') for k in ('Precision', 'Recall', 'Accuracy', 'Substitution Error', 'Miss Error',
                                                               'False Alarm Error', 'Total Error')}


def test_matching_is_maximum_not_greedy():
    """est 60.4 is admissible for refs 60 and 60.8; est 59.6 only for 60: a maximum matching pairs both."""
    from timbre_trap.utils.metrics import match_count
    assert match_count([np.array([60.0, 60.8])], [np.array([60.4, 59.6])], 0.5)[0] == 2
    rng = np.random.RandomState(3)
    for _ in range(200):                                # brute force over all injections on small random frames
        r, e = rng.uniform(59, 62, rng.randint(0, 5)), rng.uniform(59, 62, rng.randint(0, 5))
        best = 0
        adj = np.abs(np.subtract.outer(r, e)) <= 0.5
        small, big, a = (r, e, adj) if len(r) <= len(e) else (e, r, adj.T)
        for perm in itertools.permutations(range(len(big)), len(small)):
            best = max(best, sum(a[i, j] for i, j in enumerate(perm)))
        assert match_count([r], [e], 0.5)[0] == best


def test_estimates_are_resampled_to_the_reference_time_base():
    from timbre_trap.utils import multipitch_metrics
    from timbre_trap.utils.metrics import resample_multipitch
    ref_t = np.array([0.0, 0.1, 0.2, 0.3])
    est_t = np.array([0.04, 0.16, 0.26])                 # nearest: 0.0 -> out of range (empty), 0.1 -> tie 0.04/0.16 -> earlier
    est = [_hz([60]), _hz([62]), _hz([64])]
    rs = resample_multipitch(est_t, est, ref_t)
    assert [len(f) for f in rs] == [0, 1, 1, 0]
    np.testing.assert_allclose(rs[1], _hz([60]))
    np.testing.assert_allclose(rs[2], _hz([62]))         # 0.2 is nearer to 0.16 than to 0.26
    ref = [_hz([60]), _hz([60]), _hz([62]), _hz([64])]
    s = multipitch_metrics(ref_t, ref, est_t, est)
    assert s['Precision'] == pytest.approx(1.0) and s['Recall'] == pytest.approx(2 / 4)
    with pytest.raises(ValueError):
        multipitch_metrics(ref_t, ref, est_t, [np.array([6000.0])] * 3)      # above mir_eval's 5 kHz limit


def test_evaluator_has_the_reference_interface():
    from timbre_trap.utils import MultipitchEvaluator
    ev = MultipitchEvaluator()
    t = np.arange(3) * 0.01
    ref = [_hz([60, 64]), _hz([62]), np.array([])]
    est = [_hz([60]), _hz([62, 70]), np.array([])]
    res = ev.evaluate(t, est, t, ref)                    # argument order of the reference: estimates first
    assert res['mpe/precision'] == pytest.approx(2 / 3) and res['mpe/recall'] == pytest.approx(2 / 3)
    pr, rc = res['mpe/precision'], res['mpe/recall']
    assert res['mpe/f1-score'] == 2 * pr * rc / (pr + rc + sys.float_info.epsilon)
    assert 'mpe/chroma total error' in res and len(res) == 15
    ev.append_results(res)
    ev.append_results({k: v / 2 for k, v in res.items()})
    mean, std = ev.average_results()
    assert mean['mpe/precision'] == round(0.75 * 2 / 3, 5) and std['mpe/precision'] == round(float(np.std([2 / 3, 1 / 3])), 5)
    ev.reset_results()
    assert ev.results == {}


def _sdr_dense(preds, target, L=512):
    """Independent derivation: least-squares projection of preds onto L delayed (zero-padded) copies of target."""
    n = len(target)
    A = np.zeros((n + L - 1, L))
    for k in range(L):
        A[k:k + n, k] = target
    y = np.concatenate((preds, np.zeros(L - 1)))
    h, *_ = np.linalg.lstsq(A, y, rcond=None)
    proj = A @ h
    return 10 * np.log10((proj ** 2).sum() / ((y - proj) ** 2).sum())


def test_sdr_against_dense_projection_and_known_answers():
    from timbre_trap.utils import signal_distortion_ratio
    rng = np.random.RandomState(0)
    n = 4000
    target = rng.randn(n)
    noise = rng.randn(n)
    # (1) white target + independent white noise at 10 dB: no 512-tap filter can explain the noise -> ~10 dB
    preds = target + noise * 10 ** (-10 / 20)
    sdr = signal_distortion_ratio(preds, target)
    assert abs(sdr - 10.0) < 1.0
    assert abs(sdr - _sdr_dense(preds, target)) < 0.05
    # (2) a filtered + delayed copy is "allowed distortion": the SDR is far above the noise-only case
    fir = np.array([0.0, 0.0, 0.0, 0.7, -0.2, 0.1])
    filt = np.convolve(target, fir)[:n]
    clean = signal_distortion_ratio(filt, target)       # finite only because the truncated convolution tail is unexplained
    assert clean > 25.0 and abs(clean - _sdr_dense(filt, target)) < 0.05
    preds2 = filt + noise * 0.05
    assert abs(signal_distortion_ratio(preds2, target) - _sdr_dense(preds2, target)) < 0.05
    # (3) scale invariance (both signals are normalised) and batching
    assert abs(signal_distortion_ratio(3.0 * preds, 0.2 * target) - sdr) < 1e-8
    both = signal_distortion_ratio(np.stack((preds, preds2)), np.stack((target, target)))
    assert both.shape == (2,) and abs(both[0] - sdr) < 1e-9
    # (4) tensors in, the reference's call shape (B x 1 x N)
    import torch
    out = signal_distortion_ratio(torch.from_numpy(preds).view(1, 1, -1).float(), torch.from_numpy(target).view(1, 1, -1).float())
    assert out.shape == (1, 1) and abs(float(out[0, 0]) - sdr) < 1e-3
