"""
Generates tests/golden/*.npz by importing the REFERENCE (read-only at /root/reference) in this
container and recording its outputs on closed-form inputs and weights.  Run once here:

    python tests/golden/make_golden.py

The reference itself never travels: only these inputs/outputs do.  Three third-party modules
that the reference imports are absent from the image and are stubbed *for import only*:
``torchaudio.transforms.AmplitudeToDB`` (unused by every fixture), ``librosa.hz_to_midi``
(the documented one-line formula) and ``cqt_pytorch.CQT`` (replaced by tests/golden/stub_cqt.py,
so no fixture pins CQT *values* -- see oracle/nsgt.py header: parity unpinned).
"""

import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import stub_cqt  # noqa: E402
from oracle import autoencoder as oae  # noqa: E402  (closed-form weights + shape helper only)

STUB_BLOCK, STUB_M = 64, 16


def install_stubs():
    ta = types.ModuleType('torchaudio')
    tat = types.ModuleType('torchaudio.transforms')

    class AmplitudeToDB(nn.Module):
        def __init__(self, stype='power', top_db=None):
            super().__init__()
    tat.AmplitudeToDB = AmplitudeToDB
    ta.transforms = tat
    sys.modules['torchaudio'] = ta
    sys.modules['torchaudio.transforms'] = tat

    cp = types.ModuleType('cqt_pytorch')

    class CQT(nn.Module):
        def __init__(self, num_octaves, num_bins_per_octave, sample_rate, block_length, power_of_2_length=False):
            super().__init__()
            self._n_bins = num_octaves * num_bins_per_octave
            self.block_length = block_length
            self.max_window_length = 1024 if block_length == 66150 else STUB_M

        def encode(self, audio):
            return stub_cqt.stub_encode(audio, self._n_bins, self.block_length, self.max_window_length)
    cp.CQT = CQT
    sys.modules['cqt_pytorch'] = cp

    lb = types.ModuleType('librosa')
    lb.hz_to_midi = lambda f: 12 * (np.log2(np.asanyarray(f)) - np.log2(440.0)) + 69
    lb.midi_to_hz = lambda m: 440.0 * (2.0 ** ((np.asanyarray(m) - 69.0) / 12.0))
    sys.modules['librosa'] = lb


def npy(t):
    return t.detach().cpu().numpy()


def load_closed_form(module, prefix_filter=None, **kw):
    """Fill a reference module with closed-form weights keyed by ITS OWN state_dict order."""
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = oae.closed_form_state_dict(shapes, **kw)
    module.load_state_dict(sd)
    return sd


def main():
    install_stubs()
    sys.path.insert(0, '/root/reference')
    import timbre_trap.framework as ref
    from timbre_trap.framework import objectives as robj
    torch.manual_seed(0)
    torch.set_grad_enabled(True)

    # ---- 1. blocks --------------------------------------------------------------------------
    out = {}
    x = stub_cqt.closed_form_coefficients(1, 12, 10)[:, :1].repeat(1, 4, 1, 1) * \
        torch.tensor([1.0, -0.5, 0.25, 2.0]).view(1, 4, 1, 1)
    out['res_x'] = npy(x)
    for d in (1, 2, 3):
        m = ref.ResidualConv2dBlock(4, 4, kernel_size=3, dilation=d)
        sd = load_closed_form(m)
        out[f'res_d{d}_y'] = npy(m(x))
        for k, v in sd.items():
            out[f'res_d{d}_sd.{k}'] = npy(v)

    xe = stub_cqt.closed_form_coefficients(2, 13, 6)               # (2,2,13,6)
    m = ref.EncoderBlock(2, 4)
    sd = load_closed_form(m)
    out['encblk_x'] = npy(xe)
    out['encblk_y'] = npy(m(xe))
    for k, v in sd.items():
        out[f'encblk_sd.{k}'] = npy(v)
    xd = stub_cqt.closed_form_coefficients(2, 5, 6).repeat(1, 2, 1, 1)   # (2,4,5,6)
    out['decblk_x'] = npy(xd)
    for p in (0, 1):
        m = ref.DecoderBlock(4, 2, padding=p)
        sd = load_closed_form(m)
        out[f'decblk_p{p}_y'] = npy(m(xd))
        for k, v in sd.items():
            out[f'decblk_p{p}_sd.{k}'] = npy(v)
    np.savez_compressed(os.path.join(HERE, 'blocks.npz'), **out)

    # ---- 2. encoder / decoder at F=540 -------------------------------------------------------
    out = {}
    for mc, lat in ((1, None), (2, 128)):
        enc = ref.Encoder(540, lat, mc)
        dec = ref.Decoder(540, lat, mc)
        load_closed_form(enc)
        load_closed_form(dec)
        coeffs = stub_cqt.closed_form_coefficients(1, 540, 6)
        latents, emb, _ = enc(coeffs)
        out[f'mc{mc}_latents'] = npy(latents)
        for i, e in enumerate(emb):
            out[f'mc{mc}_emb{i}'] = npy(e)
        ind = torch.ones_like(latents[..., :1, :])
        out[f'mc{mc}_dec'] = npy(dec(torch.cat((latents, ind), -2)))
        out[f'mc{mc}_dec_skip'] = npy(dec(torch.cat((latents, 0 * ind), -2), emb))
    np.savez_compressed(os.path.join(HERE, 'encdec.npz'), **out)

    # ---- 3. TimbreTrap.forward / inference / chunked_inference with the stub transform -------
    out = {}
    secs = (STUB_BLOCK + 0.5) / 22050
    for tag, kw in (('mc1', dict(model_complexity=1)),
                    ('mc2skip', dict(model_complexity=2, latent_size=128, skip_connections=True))):
        model = ref.TimbreTrap(22050, 9, 60, secs, **kw)
        assert model.sliCQ.block_length == STUB_BLOCK
        load_closed_form(model)
        audio = stub_cqt.closed_form_audio(2, STUB_BLOCK)
        res = model(audio, consistency=True)
        for name, t in zip(('reconstruction', 'latents', 'transcription', 'transcription_rec', 'transcription_scr'), res[:5]):
            out[f'{tag}_fwd_{name}'] = npy(t)
        out[f'{tag}_act'] = npy(model.to_activations(res[2]))
        res_nc = model(audio, consistency=False)
        assert res_nc[3] is None and res_nc[4] is None and res_nc[5] == {}
        long_audio = stub_cqt.closed_form_audio(1, int(2.5 * STUB_BLOCK))
        model.eval()
        out[f'{tag}_chunked_trn'] = npy(model.chunked_inference(long_audio, True))
        out[f'{tag}_chunked_rec'] = npy(model.chunked_inference(long_audio, False))
        out[f'{tag}_inference'] = npy(model.inference(long_audio, False))
        out[f'{tag}_transcribe'] = npy(model.transcribe(long_audio))
        model.train()

        # gradients of the train.py total loss w.r.t. every parameter (tiny case)
        coeffs = model.sliCQ(audio)
        gt = stub_cqt.closed_form_targets(2, 540, STUB_M)
        rec, lat, trn, trn_rec, trn_scr, _ = model(audio, True)
        act = model.to_activations(trn)
        l_rec = robj.compute_reconstruction_loss(rec, coeffs)
        l_trn = robj.compute_transcription_loss(act, gt, True)
        l_sp, l_sc = robj.compute_consistency_loss(trn_rec, trn_scr, trn)
        total = l_rec + l_trn + (l_sp + l_sc)
        model.zero_grad()
        total.backward()
        out[f'{tag}_losses'] = np.array([float(l_rec), float(l_trn), float(l_sp), float(l_sc), float(total)], dtype=np.float64)
        for k, p in model.named_parameters():
            g = p.grad.double().flatten()
            if tag == 'mc1':
                out[f'{tag}_grad.{k}'] = npy(p.grad)
            else:   # keep the fixture small: [sum, L2 norm, first 6 values]
                out[f'{tag}_gradstat.{k}'] = npy(torch.cat([g.sum().view(1), g.norm().view(1), g[:6]]))
    np.savez_compressed(os.path.join(HERE, 'model.npz'), **out)

    # ---- 4. objectives ----------------------------------------------------------------------
    out = {}
    a = stub_cqt.closed_form_coefficients(2, 540, 5).requires_grad_(True)
    b = (stub_cqt.closed_form_coefficients(2, 540, 5) * 0.7 + 0.1).flip(-1).requires_grad_(True)
    l = robj.compute_reconstruction_loss(a, b)
    ga, gb = torch.autograd.grad(l, (a, b))
    out['rec_loss'] = np.float64(float(l))
    out['rec_ga_sum'] = npy(ga.abs().sum())
    out['rec_ga'] = npy(ga[:, :, ::45])
    out['rec_gb'] = npy(gb[:, :, ::45])
    est = torch.sigmoid(stub_cqt.closed_form_coefficients(2, 540, 7)[:, 0]).requires_grad_(True)
    tgt = stub_cqt.closed_form_targets(2, 540, 7)
    assert (tgt == 1).any() and (tgt.sum(-2) == 0).any()
    for w in (False, True):
        l = robj.compute_transcription_loss(est, tgt, w)
        g, = torch.autograd.grad(l, est)
        out[f'trn_loss_w{int(w)}'] = np.float64(float(l))
        out[f'trn_grad_w{int(w)}'] = npy(g)
    tgt_ones = torch.ones(1, 540, 2)                  # neg == 0 -> zero scale -> forced to 1
    l = robj.compute_transcription_loss(est[:1, :, :2], tgt_ones, True)
    out['trn_loss_allones'] = np.float64(float(l))
    sp, sc = robj.compute_consistency_loss(a, b, (a + b) / 2)
    out['cons'] = np.array([float(sp), float(sc)])
    np.savez_compressed(os.path.join(HERE, 'objectives.npz'), **out)

    # ---- 5. wrapper arithmetic (cqtwrapper.py a5 helpers) on the real block length -----------
    out = {}
    cq = ref.CQT(9, 60, 22050, 3)
    assert cq.block_length == 66150 and cq.max_window_length == 1024
    out['hop_length'] = np.float64(cq.hop_length)
    out['midi_freqs'] = cq.get_midi_freqs()
    ns = np.array([0, 1, 64, 65, 66149, 66150, 66151, 100000, 132300, 165375, 198450, 1234567], dtype=np.int64)
    out['frames_in'] = ns
    out['frames_out'] = np.array([cq.get_expected_frames(int(n)) for n in ns], dtype=np.int64)
    ts = np.array([-1.0, 0.0, 0.5, 2.9999, 3.0, 3.00001, 9.0, 12.345], dtype=np.float64)
    out['samples_in'] = ts
    out['samples_out'] = np.array([cq.get_expected_samples(float(t)) for t in ts], dtype=np.int64)
    out['times_3100'] = cq.get_times(3100)
    out['pad_lens_in'] = ns
    out['pad_lens_out'] = np.array([cq.pad_to_block_length(torch.zeros(1, 1, int(n))).size(-1) for n in ns], dtype=np.int64)
    c = stub_cqt.stub_encode(stub_cqt.closed_form_audio(2, STUB_BLOCK), 540, STUB_BLOCK, STUB_M)
    r = ref.CQT.to_real(c)
    out['to_real_in_re'], out['to_real_in_im'] = npy(c.real), npy(c.imag)
    out['to_real_out'] = npy(r)
    out['to_magnitude_out'] = npy(ref.CQT.to_magnitude(r))
    cc = ref.CQT.to_complex(r)
    out['to_complex_re'], out['to_complex_im'] = npy(cc.real), npy(cc.imag)
    np.savez_compressed(os.path.join(HERE, 'wrapper.npz'), **out)

    for fn in sorted(os.listdir(HERE)):
        if fn.endswith('.npz'):
            print(fn, os.path.getsize(os.path.join(HERE, fn)))




def make_utils_golden():
    """f1 / f3 rows: CosineWarmup learning rates, gradient-norm helpers, filter_non_peaks / threshold from the reference utils."""
    install_stubs()
    me = types.ModuleType('mir_eval')
    sys.modules.setdefault('mir_eval', me)
    sys.path.insert(0, '/root/reference')
    from timbre_trap.utils import experiments as rex, processing as rpr
    out = {}
    lin = torch.nn.Linear(3, 2)
    for n_steps in (0, 1, 7, 50):
        opt = torch.optim.AdamW(lin.parameters(), lr=1e-3)
        sch = rex.CosineWarmup(opt, n_steps=n_steps)
        lrs, active = [opt.param_groups[0]['lr']], [sch.is_active()]
        for _ in range(n_steps + 3):
            opt.step()
            sch.step()
            lrs.append(opt.param_groups[0]['lr'])
            active.append(sch.is_active())
        out[f'warmup_lr_{n_steps}'] = np.array(lrs, dtype=np.float64)
        out[f'warmup_active_{n_steps}'] = np.array(active, dtype=np.int64)
    # gradient statistics on a closed-form module
    torch.manual_seed(0)
    mod = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.Linear(4, 3))
    i = 0
    for p in mod.parameters():
        n = p.numel()
        p.grad = (torch.sin(0.7 * torch.arange(n, dtype=torch.float32) + i) * (i + 1)).view_as(p)
        i += 1
    out['grad_stats'] = np.array([rex.sum_gradient_norms(mod), rex.average_gradient_norms(mod), rex.get_max_gradient(mod),
                                  rex.get_max_gradient_norm(mod)], dtype=np.float64)
    # post-processing
    a = stub_cqt.closed_form_coefficients(2, 40, 9)[:, 0].numpy().astype(np.float64)          # (2, 40, 9)
    a = np.abs(a)
    a[0, 0, 0] = 3.0            # edge peak at the first row
    a[1, -1, 2] = 3.0           # edge peak at the last row
    a[0, 10:13, 4] = 1.5        # plateau: no strict maximum
    out['pp_in'] = a
    out['pp_peaks'] = rpr.filter_non_peaks(a)
    out['pp_thr'] = rpr.threshold(a, 0.6)
    out['pp_peaks_thr'] = rpr.threshold(rpr.filter_non_peaks(a), 0.3)
    np.savez_compressed(os.path.join(HERE, 'utils.npz'), **out)
    print('utils.npz', os.path.getsize(os.path.join(HERE, 'utils.npz')))


def closed_form_multi_pitch(midi_freqs, T=48):
    """Ragged frame-level pitch lists (Hz) exercising the corner cases of the target generation."""
    hz = lambda m: 440.0 * (2.0 ** ((np.asarray(m, dtype=np.float64) - 69.0) / 12.0))
    mp = []
    for t in range(T):
        if t % 7 == 3:
            mp.append(np.empty(0))                                    # silent frame
            continue
        k = (37 * t) % 500 + 10
        p = [hz(midi_freqs[k] + 0.03 * ((t % 5) - 2))]               # slightly detuned from a bin centre
        if t % 3 == 0:
            p.append(hz(midi_freqs[(k + 1) % 540] - 0.02))            # neighbouring bin: blurs overlap and clip at 1
        if t % 4 == 1:
            p.append(hz(0.5 * (midi_freqs[k + 20] + midi_freqs[k + 21])))   # exactly between two bins (tie)
        if t % 5 == 2:
            p.append(0.0)                                             # "no pitch" marker, filtered
        if t == 12:
            p += [hz(midi_freqs[-1] + 1.0), hz(midi_freqs[0] - 0.5)]  # outside the bin range: dropped with a warning
        if t == 11:
            p += [hz(midi_freqs[0]), hz(midi_freqs[-1])]              # the edge bins
        mp.append(np.array(p, dtype=np.float64))
    return mp


def make_targets_golden():
    """f3 / f4 rows: PitchDataset.multi_pitch_to_activations / activations_to_multi_pitch of the reference."""
    import warnings
    install_stubs()
    sys.modules.setdefault('mir_eval', types.ModuleType('mir_eval'))
    sys.path.insert(0, '/root/reference')
    from timbre_trap.datasets import PitchDataset
    from timbre_trap.framework import CQT
    midi_freqs = CQT(n_octaves=9, bins_per_octave=60, sample_rate=22050, secs_per_block=3).midi_freqs
    mp = closed_form_multi_pitch(midi_freqs)
    out = {'midi_freqs': np.asarray(midi_freqs, dtype=np.float64),
           'mp_values': np.concatenate(mp), 'mp_counts': np.array([len(p) for p in mp], dtype=np.int64)}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        out['act_blur'] = PitchDataset.multi_pitch_to_activations(mp, midi_freqs, 2.5)
        out['act_noblur'] = PitchDataset.multi_pitch_to_activations(mp, midi_freqs, 0)
        out['act_blur_wide'] = PitchDataset.multi_pitch_to_activations(mp, midi_freqs, 5.0)
    # inverse direction on float32 network-like activations
    a = np.abs(stub_cqt.closed_form_coefficients(1, 540, 48)[0, 0].numpy()).astype(np.float32)
    a = np.tanh(a)
    a[100, 5] = 0.7; a[101, 5] = 0.7                                  # plateau: no strict peak
    a[0, 6] = 0.9; a[539, 7] = 0.95                                   # peaks on the edge rows
    a[:, 8] = 0.0                                                     # silent frame
    out['a2mp_in'] = a
    for tag, kw in (('plain', dict(peaks_only=False, t=0.5)), ('peaks', dict(peaks_only=True, t=0.5)), ('peaks07', dict(peaks_only=True, t=0.7))):
        res = PitchDataset.activations_to_multi_pitch(a, midi_freqs, **kw)
        out['a2mp_%s_values' % tag] = np.concatenate([np.asarray(r, dtype=np.float64) for r in res]) if len(res) else np.empty(0)
        out['a2mp_%s_counts' % tag] = np.array([len(r) for r in res], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, 'targets.npz'), **out)
    print('targets.npz', os.path.getsize(os.path.join(HERE, 'targets.npz')))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'utils':
    make_utils_golden()

if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'targets':
    make_targets_golden()



def make_slicing_golden():
    """f4 row: AudioDataset.slice_audio, PitchDataset.slice_times / resample_multi_pitch of the reference (index arithmetic)."""
    install_stubs()
    for name in ('mir_eval', 'jams', 'mido'):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.path.insert(0, '/root/reference')
    from timbre_trap.datasets import AudioDataset, PitchDataset
    from timbre_trap.framework import CQT
    cqt = CQT(n_octaves=9, bins_per_octave=60, sample_rate=22050, secs_per_block=3)
    out = {}

    class Fake:
        pass

    def fake(seed, n_secs=3):
        f = Fake()
        f.cqt, f.n_secs, f.sample_rate, f.rng, f.resample_idcs = cqt, n_secs, 22050, np.random.RandomState(seed), [0, -1]
        return f
    # slice_audio: long track (random and fixed offsets), short track (random and fixed padding), explicit n_samples
    long_audio = torch.arange(200000, dtype=torch.float32).view(1, -1) / 200000
    short_audio = torch.arange(30000, dtype=torch.float32).view(1, -1) / 30000
    cases = [('long_rand', long_audio, None, None, 5), ('long_fixed', long_audio, None, 12345, 5), ('short_rand', short_audio, None, None, 6),
             ('short_fixed', short_audio, None, -4000, 6), ('explicit_n', long_audio, 50000, None, 7), ('exact', long_audio[:, :66150], None, None, 8)]
    for tag, audio, n_samples, offset_s, seed in cases:
        a, off = AudioDataset.slice_audio(fake(seed), audio, n_samples, offset_s)
        out['sa_%s_first_last_sum' % tag] = np.array([float(a[0, 0]), float(a[0, -1]), float(a.double().sum()), a.size(-1)], dtype=np.float64)
        out['sa_%s_offset' % tag] = np.array(off, dtype=np.float64)
    # slice_times
    t_long = cqt.get_times(5000)
    t_short = cqt.get_times(700)
    for tag, times, n_frames, offset_t, seed in [('long_rand', t_long, None, None, 3), ('long_offset', t_long, None, 1.2345, 3),
                                                 ('short_rand', t_short, None, None, 4), ('short_offset', t_short, None, -0.25, 4),
                                                 ('explicit', t_long, 333, None, 9)]:
        ts, off = PitchDataset.slice_times(fake(seed), times, n_frames, offset_t)
        out['st_%s_times' % tag] = np.asarray(ts, dtype=np.float64)
        out['st_%s_offset' % tag] = np.array(off, dtype=np.float64)
    # resample_multi_pitch: irregular source grid, targets with ties, outside the span, and +-inf padding
    src_t = np.cumsum(np.array([0.0, 0.01, 0.02, 0.005, 0.03, 0.01, 0.0125, 0.02, 0.01, 0.015]))
    src_mp = [np.array([100.0 + 10 * i, 200.0 + i]) if i % 3 else np.empty(0) for i in range(len(src_t))]
    tgt = np.concatenate(([-np.inf, -1.0], src_t[:-1] + np.diff(src_t) / 2, src_t, [src_t[3] + 1e-9, 0.5, np.inf]))
    for tag, idcs in (('default', [0, -1]), ('inner', [1, -2])):
        f = fake(0)
        f.resample_idcs = idcs
        res = PitchDataset.resample_multi_pitch(f, src_t, src_mp, tgt)
        out['rs_%s_values' % tag] = np.concatenate([np.asarray(r, dtype=np.float64) for r in res])
        out['rs_%s_counts' % tag] = np.array([len(r) for r in res], dtype=np.int64)
    out['rs_src_t'], out['rs_tgt'] = src_t, tgt
    np.savez_compressed(os.path.join(HERE, 'slicing.npz'), **out)
    print('slicing.npz', os.path.getsize(os.path.join(HERE, 'slicing.npz')))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'slicing':
    make_slicing_golden()


if __name__ == '__main__' and len(sys.argv) == 1:
    main()
    make_utils_golden()
    make_targets_golden()
    make_slicing_golden()
