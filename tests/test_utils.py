"""train-loop helpers and post-processing vs the reference-generated golden (tests/golden/utils.npz)."""

import os

import numpy as np
import pytest
import torch

from oracle import postprocessing as opp

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'utils.npz'))


def _closed_form_module(device='cpu'):
    # same construction as tests/golden/make_golden.py:make_utils_golden
    mod = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.Linear(4, 3)).to(device)
    for i, p in enumerate(mod.parameters()):
        p.grad = (torch.sin(0.7 * torch.arange(p.numel(), dtype=torch.float32) + i) * (i + 1)).view_as(p).to(device)
    return mod


# ---------------------------------------------------------------- CPU: oracle pinned + host logic
@pytest.mark.parametrize('n_steps', [0, 1, 7, 50])
def test_oracle_cosine_warmup_golden(n_steps):
    want = GOLD[f'warmup_lr_{n_steps}']
    got = np.array([1e-3 * opp.cosine_warmup_scale(i, n_steps) for i in range(len(want))])
    np.testing.assert_allclose(got, want, rtol=1e-14, atol=0)


def test_oracle_postprocessing_golden():
    a = GOLD['pp_in']
    assert np.array_equal(opp.filter_non_peaks(a), GOLD['pp_peaks'])
    assert np.array_equal(opp.threshold(a, 0.6), GOLD['pp_thr'])
    assert np.array_equal(opp.threshold(opp.filter_non_peaks(a), 0.3), GOLD['pp_peaks_thr'])
    # the fixture holds the edge cases: peaks on the first / last row survive, a plateau does not
    assert GOLD['pp_peaks'][0, 0, 0] == 3.0 and GOLD['pp_peaks'][1, -1, 2] == 3.0
    assert not GOLD['pp_peaks'][0, 10:13, 4].any()


def test_oracle_gradient_statistics_golden():
    mod = _closed_form_module()
    got = opp.gradient_statistics([p.grad.numpy() for p in mod.parameters()])
    np.testing.assert_allclose(got, GOLD['grad_stats'], rtol=1e-6)


@pytest.mark.parametrize('n_steps', [0, 1, 7, 50])
def test_cosine_warmup_matches_reference(n_steps):
    from timbre_trap.utils import CosineWarmup
    lin = torch.nn.Linear(3, 2)
    opt = torch.optim.AdamW(lin.parameters(), lr=1e-3)
    sch = CosineWarmup(opt, n_steps=n_steps)
    lrs, active = [opt.param_groups[0]['lr']], [sch.is_active()]
    for _ in range(n_steps + 3):
        opt.step()
        sch.step()
        lrs.append(opt.param_groups[0]['lr'])
        active.append(sch.is_active())
    np.testing.assert_allclose(lrs, GOLD[f'warmup_lr_{n_steps}'], rtol=1e-14, atol=0)
    assert np.array_equal(np.array(active, dtype=np.int64), GOLD[f'warmup_active_{n_steps}'])
    sch.reset()
    assert opt.param_groups[0]['lr'] == pytest.approx(GOLD[f'warmup_lr_{n_steps}'][0], rel=1e-14)


def test_helpers_refuse_cpu_tensors():
    from timbre_trap.utils import filter_non_peaks, sum_gradient_norms
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        filter_non_peaks(torch.zeros(4, 4))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        sum_gradient_norms(_closed_form_module())


# ---------------------------------------------------------------- GPU: kernels vs golden / oracle
@pytest.mark.gpu
def test_gpu_postprocessing_golden():
    from timbre_trap.utils import filter_non_peaks, threshold, peaks_above
    a = GOLD['pp_in']
    out = filter_non_peaks(a)                     # ndarray in -> float64 ndarray out, like the reference
    assert out.dtype == np.float64 and np.array_equal(out, GOLD['pp_peaks'])
    assert np.array_equal(threshold(a, 0.6), GOLD['pp_thr'])
    assert np.array_equal(threshold(filter_non_peaks(a), 0.3), GOLD['pp_peaks_thr'])
    t = torch.from_numpy(a).float().cuda()
    assert torch.equal(filter_non_peaks(t).cpu().double(), torch.from_numpy(GOLD['pp_peaks']))
    assert torch.equal(peaks_above(t, 0.3).cpu().double(), torch.from_numpy(GOLD['pp_peaks_thr']))


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(1, 1), (1, 7), (5, 1), (440, 1024), (3, 2, 440, 257), (0, 440, 8)])
def test_gpu_postprocessing_oracle(shape):
    from timbre_trap.utils import filter_non_peaks, threshold, peaks_above
    g = torch.Generator().manual_seed(sum(shape))
    a = torch.rand(shape, generator=g)
    a[a < 0.2] = 0.0                              # ties with the zero padding rows / with each other
    want = opp.filter_non_peaks(a.numpy())
    assert np.array_equal(filter_non_peaks(a.cuda()).cpu().numpy().astype(np.float64), want)
    assert np.array_equal(threshold(a.cuda(), 0.5).cpu().numpy(), opp.threshold(a.numpy(), 0.5))
    assert np.array_equal(peaks_above(a.cuda(), 0.5).cpu().numpy(), opp.threshold(want, 0.5))


@pytest.mark.gpu
def test_gpu_gradient_statistics_golden():
    from timbre_trap.utils import (sum_gradient_norms, average_gradient_norms, get_max_gradient, get_max_gradient_norm)
    mod = _closed_form_module('cuda')
    got = [sum_gradient_norms(mod), average_gradient_norms(mod), get_max_gradient(mod), get_max_gradient_norm(mod)]
    np.testing.assert_allclose(got, GOLD['grad_stats'], rtol=1e-6)


@pytest.mark.gpu
def test_gpu_gradient_statistics_flat_buffer():
    """Gradients that are views of FusedAdamW's flat buffer are read in place; same numbers as the oracle."""
    from timbre_trap.framework import TimbreTrap
    from timbre_trap.utils import FusedAdamW, gradient_statistics
    torch.manual_seed(3)
    model = TimbreTrap(sample_rate=22050, n_octaves=9, bins_per_octave=60, latent_size=16, model_complexity=1).cuda()
    opt = FusedAdamW(model.parameters(), lr=1e-3)
    opt.flat_grad.copy_(torch.randn(opt.flat_grad.shape, generator=torch.Generator().manual_seed(5)).cuda())
    norms, amax = gradient_statistics(model)
    want = [opp.gradient_statistics([p.grad.cpu().numpy()]) for _, p in model.named_parameters()]
    np.testing.assert_allclose(norms, [w[0] for w in want], rtol=2e-6)
    np.testing.assert_allclose(amax, [w[2] for w in want], rtol=0, atol=0)


# ---------------------------------------------------------------- f3 / f4: target generation and its inverse
TGT = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'targets.npz'))


def _ragged(values, counts):
    return np.split(values, np.cumsum(counts)[:-1])


@pytest.mark.parametrize('key,decay', [('act_blur', 2.5), ('act_noblur', 0), ('act_blur_wide', 5.0)])
def test_oracle_multi_pitch_to_activations_golden(key, decay):
    from oracle import targets as ot
    got = ot.multi_pitch_to_activations(_ragged(TGT['mp_values'], TGT['mp_counts']), TGT['midi_freqs'], decay)
    assert np.array_equal(got, TGT[key])                       # bit-exact float64, incl. the SciPy blur order
    assert (TGT[key] == 1.0).sum() >= 60 and TGT[key].max() == 1.0


@pytest.mark.parametrize('tag,kw', [('plain', dict(peaks_only=False, t=0.5)), ('peaks', dict(peaks_only=True, t=0.5)),
                                    ('peaks07', dict(peaks_only=True, t=0.7))])
def test_oracle_activations_to_multi_pitch_golden(tag, kw):
    from oracle import targets as ot
    res = ot.activations_to_multi_pitch(TGT['a2mp_in'], TGT['midi_freqs'], **kw)
    assert np.array_equal(np.array([len(r) for r in res]), TGT['a2mp_%s_counts' % tag])
    assert np.array_equal(np.concatenate(res), TGT['a2mp_%s_values' % tag])


def test_target_helpers_refuse_cpu():
    from timbre_trap.utils import multi_pitch_to_activations
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        multi_pitch_to_activations([np.array([440.0])], TGT['midi_freqs'], device='cpu')


@pytest.mark.gpu
@pytest.mark.parametrize('key,decay', [('act_blur', 2.5), ('act_noblur', 0), ('act_blur_wide', 5.0)])
def test_gpu_multi_pitch_to_activations_golden(key, decay):
    import warnings
    from timbre_trap.datasets import PitchDataset
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        got = PitchDataset.multi_pitch_to_activations(_ragged(TGT['mp_values'], TGT['mp_counts']), TGT['midi_freqs'], decay)
    assert any('Could not fully represent' in str(x.message) for x in w)       # the out-of-range pitches of frame 12
    assert got.dtype == np.float64 and got.shape == TGT[key].shape
    assert np.array_equal(got, TGT[key])                       # bit-exact against the reference's SciPy arithmetic


@pytest.mark.gpu
def test_gpu_multi_pitch_to_activations_edge_cases():
    from oracle import targets as ot
    from timbre_trap.utils import multi_pitch_to_activations
    mf = TGT['midi_freqs']
    assert not multi_pitch_to_activations([np.empty(0)] * 5, mf).any()                      # all silent
    assert multi_pitch_to_activations([], mf).shape == (540, 0)                              # no frames
    g = np.random.default_rng(3)
    mp = [ot.midi_to_hz(g.uniform(mf[0], mf[-1], size=g.integers(0, 6))) for _ in range(1500)]
    got = multi_pitch_to_activations(mp, mf, return_tensor=True)
    assert got.is_cuda and got.dtype == torch.float64
    assert np.array_equal(got.cpu().numpy(), ot.multi_pitch_to_activations(mp, mf))


@pytest.mark.gpu
@pytest.mark.parametrize('tag,kw', [('plain', dict(peaks_only=False, t=0.5)), ('peaks', dict(peaks_only=True, t=0.5)),
                                    ('peaks07', dict(peaks_only=True, t=0.7))])
def test_gpu_activations_to_multi_pitch_golden(tag, kw):
    from timbre_trap.datasets import PitchDataset
    for a in (TGT['a2mp_in'], torch.from_numpy(TGT['a2mp_in']).cuda()):
        res = PitchDataset.activations_to_multi_pitch(a, TGT['midi_freqs'], **kw)
        assert np.array_equal(np.array([len(r) for r in res]), TGT['a2mp_%s_counts' % tag])
        assert np.array_equal(np.concatenate(res), TGT['a2mp_%s_values' % tag])


@pytest.mark.gpu
def test_gpu_peaks_above_with_bin_mask():
    """The evaluate()-side composition: zero the bins above the scoring range, keep strict peaks, threshold."""
    from timbre_trap.utils import peaks_above
    g = torch.Generator().manual_seed(11)
    a = torch.rand(2, 540, 300, generator=g)
    masked = a.clone()
    masked[:, 472:] = 0
    want = opp.threshold(opp.filter_non_peaks(masked.numpy()), 0.6)
    got = peaks_above(a.cuda(), 0.6, n_valid_bins=472).cpu().numpy()
    assert np.array_equal(got, want)
    assert not got[:, 472:].any() and got[:, 471].any()          # row 471 now compares against a zero row above it
