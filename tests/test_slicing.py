"""
f4 row: excerpt slicing and annotation resampling (timbre_trap/utils/slicing.py) against values recorded from the imported
reference (tests/golden/slicing.npz, generator tests/golden/make_golden.py slicing) and against SciPy's interp1d directly.
"""

import numpy as np
import scipy.interpolate
import torch


def _cqt():
    from timbre_trap.framework import CQT
    return CQT(n_octaves=9, bins_per_octave=60, sample_rate=22050, secs_per_block=3)


def test_slice_audio_matches_reference(golden):
    from timbre_trap.utils import ExcerptSlicer
    g = golden('slicing')
    cqt = _cqt()
    long_audio = torch.arange(200000, dtype=torch.float32).view(1, -1) / 200000
    short_audio = torch.arange(30000, dtype=torch.float32).view(1, -1) / 30000
    cases = [('long_rand', long_audio, None, None, 5), ('long_fixed', long_audio, None, 12345, 5), ('short_rand', short_audio, None, None, 6),
             ('short_fixed', short_audio, None, -4000, 6), ('explicit_n', long_audio, 50000, None, 7), ('exact', long_audio[:, :66150], None, None, 8)]
    for tag, audio, n_samples, offset_s, seed in cases:
        s = ExcerptSlicer(cqt, n_secs=3, sample_rate=22050, seed=seed)
        a, off = s.slice_audio(audio, n_samples, offset_s)
        got = np.array([float(a[0, 0]), float(a[0, -1]), float(a.double().sum()), a.size(-1)])
        np.testing.assert_array_equal(got, g['sa_%s_first_last_sum' % tag], err_msg=tag)
        assert off == float(g['sa_%s_offset' % tag]), tag            # bit-exact offsets (float64 division)


def test_slice_times_matches_reference(golden):
    from timbre_trap.utils import ExcerptSlicer
    g = golden('slicing')
    cqt = _cqt()
    t_long, t_short = cqt.get_times(5000), cqt.get_times(700)
    for tag, times, n_frames, offset_t, seed in [('long_rand', t_long, None, None, 3), ('long_offset', t_long, None, 1.2345, 3),
                                                 ('short_rand', t_short, None, None, 4), ('short_offset', t_short, None, -0.25, 4),
                                                 ('explicit', t_long, 333, None, 9)]:
        s = ExcerptSlicer(cqt, n_secs=3, sample_rate=22050, seed=seed)
        ts, off = s.slice_times(times, n_frames, offset_t)
        np.testing.assert_array_equal(np.asarray(ts, dtype=np.float64), g['st_%s_times' % tag], err_msg=tag)
        assert float(off) == float(g['st_%s_offset' % tag]), tag


def test_resample_multi_pitch_matches_reference(golden):
    from timbre_trap.utils import resample_multi_pitch
    g = golden('slicing')
    src_t, tgt = g['rs_src_t'], g['rs_tgt']
    src_mp = [np.array([100.0 + 10 * i, 200.0 + i]) if i % 3 else np.empty(0) for i in range(len(src_t))]
    for tag, idcs in (('default', [0, -1]), ('inner', [1, -2])):
        res = resample_multi_pitch(src_t, src_mp, tgt, idcs)
        np.testing.assert_array_equal(np.array([len(r) for r in res]), g['rs_%s_counts' % tag])
        np.testing.assert_array_equal(np.concatenate([np.asarray(r, dtype=np.float64) for r in res]), g['rs_%s_values' % tag])


def test_nearest_indices_equal_scipy_interp1d():
    """The nearest rule itself, on random grids with exact ties, duplicates of grid points and out-of-range targets."""
    from timbre_trap.utils import nearest_indices
    rng = np.random.RandomState(0)
    for trial in range(50):
        n = rng.randint(2, 40)
        src = np.cumsum(rng.rand(n) * rng.choice([1e-3, 1.0, 64.599609375 / 22050])) + rng.randn()
        mids = (src[1:] + src[:-1]) / 2
        tgt = np.concatenate((src, mids, src[0] - rng.rand(3), src[-1] + rng.rand(3), rng.uniform(src[0], src[-1], 20), [-np.inf, np.inf]))
        below, above = 0, n - 1
        f = scipy.interpolate.interp1d(src, np.arange(n), kind='nearest', bounds_error=False, fill_value=(below, above), assume_sorted=True)
        np.testing.assert_array_equal(nearest_indices(src, tgt, below, above), f(tgt).astype(np.int64))


def test_stand_in_dataset_classes_expose_reference_method_names():
    import timbre_trap.datasets as d
    if d.REFERENCE_DATASETS is not None:         # a reference checkout is overlaid: its own classes are in use
        return
    cqt = _cqt()
    p = d.PitchDataset(cqt, n_secs=3, seed=1)
    ts, off = p.slice_times(cqt.get_times(2000))
    assert len(ts) == 1024 and 0 <= off <= 2000 - 1024
    assert callable(d.PitchDataset.multi_pitch_to_activations) and callable(d.PitchDataset.activations_to_multi_pitch)
    a, off_t = d.AudioDataset(cqt, n_secs=3, seed=1).slice_audio(torch.zeros(1, 100000))
    assert a.shape == (1, 66150) and off_t >= 0
