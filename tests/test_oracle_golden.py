"""
Pins the oracle (oracle/autoencoder.py, oracle/objectives.py, wrapper part of oracle/nsgt.py)
against outputs of the imported reference (tests/golden/*.npz, made by tests/golden/make_golden.py).
CPU only.
"""

import numpy as np
import torch

import stub_cqt
from oracle import autoencoder as oae
from oracle import nsgt
from oracle import objectives as oobj

STUB_BLOCK, STUB_M = 64, 16
TOL = dict(rtol=2e-5, atol=2e-5)


def T(a):
    return torch.from_numpy(np.asarray(a))


def sd_from(g, prefix, rename=''):
    return {rename + k[len(prefix):]: T(g[k]) for k in g.files if k.startswith(prefix)}


def test_residual_blocks(golden):
    g = golden('blocks')
    x = T(g['res_x'])
    for d in (1, 2, 3):
        sd = sd_from(g, f'res_d{d}_sd.', 'blk.')
        y = oae.residual_block(x, sd, 'blk', d)
        np.testing.assert_allclose(y.numpy(), g[f'res_d{d}_y'], **TOL)


def test_encoder_decoder_blocks(golden):
    g = golden('blocks')
    sd = sd_from(g, 'encblk_sd.', 'e.')
    y = oae.encoder_block(T(g['encblk_x']), sd, 'e')
    assert y.shape == (2, 4, 13 // 2 - 1, 6)
    np.testing.assert_allclose(y.numpy(), g['encblk_y'], **TOL)
    for p in (0, 1):
        sd = sd_from(g, f'decblk_p{p}_sd.', 'd.')
        y = oae.decoder_block(T(g['decblk_x']), sd, 'd', p)
        assert y.shape == (2, 2, 2 * 5 + 2 + p, 6)
        np.testing.assert_allclose(y.numpy(), g[f'decblk_p{p}_y'], **TOL)


def _model_sd(mc, lat, skip=False):
    """Closed-form weights in the reference's own key order (skip_weights first)."""
    shapes = oae.state_dict_shapes(540, lat, mc, skip)
    return shapes


def test_encoder_decoder_full(golden):
    g = golden('encdec')
    for mc, lat in ((1, None), (2, 128)):
        shapes = oae.state_dict_shapes(540, lat, mc)
        enc_shapes = {k[len('encoder.'):]: v for k, v in shapes.items() if k.startswith('encoder.')}
        dec_shapes = {k[len('decoder.'):]: v for k, v in shapes.items() if k.startswith('decoder.')}
        sd = {'encoder.' + k: v for k, v in oae.closed_form_state_dict(enc_shapes).items()}
        sd.update({'decoder.' + k: v for k, v in oae.closed_form_state_dict(dec_shapes).items()})
        coeffs = stub_cqt.closed_form_coefficients(1, 540, 6)
        latents, emb = oae.encoder_forward(coeffs, sd)
        np.testing.assert_allclose(latents.numpy(), g[f'mc{mc}_latents'], **TOL)
        for i, e in enumerate(emb):
            np.testing.assert_allclose(e.numpy(), g[f'mc{mc}_emb{i}'], **TOL)
        ind = torch.ones_like(latents[..., :1, :])
        np.testing.assert_allclose(oae.decoder_forward(torch.cat((latents, ind), -2), sd).numpy(), g[f'mc{mc}_dec'], **TOL)
        np.testing.assert_allclose(oae.decoder_forward(torch.cat((latents, 0 * ind), -2), sd, emb).numpy(),
                                   g[f'mc{mc}_dec_skip'], **TOL)


def _cases():
    return (('mc1', dict(latent_size=None, model_complexity=1, skip_connections=False)),
            ('mc2skip', dict(latent_size=128, model_complexity=2, skip_connections=True)))


def _stub_forward(audio):
    c = stub_cqt.stub_encode(audio, 540, STUB_BLOCK, STUB_M)
    return torch.from_numpy(nsgt.to_real(c.numpy())).contiguous()


def test_model_forward_and_inference(golden):
    g = golden('model')
    for tag, kw in _cases():
        sd = oae.closed_form_state_dict(oae.state_dict_shapes(540, **kw))
        audio = stub_cqt.closed_form_audio(2, STUB_BLOCK)
        res = oae.forward(_stub_forward(audio), sd, consistency=True)
        for name, t in zip(('reconstruction', 'latents', 'transcription', 'transcription_rec', 'transcription_scr'), res):
            np.testing.assert_allclose(t.numpy(), g[f'{tag}_fwd_{name}'], rtol=1e-4, atol=1e-4, err_msg=f'{tag} {name}')
        np.testing.assert_allclose(oae.to_activations(res[2]).numpy(), g[f'{tag}_act'], rtol=1e-4, atol=1e-5)

        long_audio = stub_cqt.closed_form_audio(1, int(2.5 * STUB_BLOCK))
        for transcribe, key in ((True, 'chunked_trn'), (False, 'chunked_rec')):
            o = oae.chunked_inference(long_audio, sd, _stub_forward, STUB_BLOCK, STUB_M, transcribe)
            assert o.shape == g[f'{tag}_{key}'].shape == (1, 2, 540, 3 * STUB_M)
            np.testing.assert_allclose(o.numpy(), g[f'{tag}_{key}'], rtol=1e-4, atol=1e-4)
        padded = torch.from_numpy(nsgt.pad_to_block_length(long_audio.numpy(), STUB_BLOCK))
        o = oae.inference_coefficients(_stub_forward(padded), sd, False)
        np.testing.assert_allclose(o.numpy(), g[f'{tag}_inference'], rtol=1e-4, atol=1e-4)
        o = oae.to_activations(oae.chunked_inference(long_audio, sd, _stub_forward, STUB_BLOCK, STUB_M, True))
        np.testing.assert_allclose(o.numpy(), g[f'{tag}_transcribe'], rtol=1e-4, atol=1e-5)


def test_model_gradients(golden):
    g = golden('model')
    for tag, kw in _cases():
        sd = oae.closed_form_state_dict(oae.state_dict_shapes(540, **kw))
        params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        audio = stub_cqt.closed_form_audio(2, STUB_BLOCK)
        coeffs = _stub_forward(audio)
        gt = stub_cqt.closed_form_targets(2, 540, STUB_M)
        outputs = oae.forward(coeffs, params, consistency=True)
        total, parts = oobj.total_loss(outputs, coeffs, gt)
        ref = g[f'{tag}_losses']
        got = [float(parts[k]) for k in ('reconstruction', 'transcription', 'consistency_spectral', 'consistency_score', 'total')]
        np.testing.assert_allclose(got, ref, rtol=2e-5)
        total.backward()
        for k, p in params.items():
            if tag == 'mc1':
                np.testing.assert_allclose(p.grad.numpy(), g[f'{tag}_grad.{k}'], rtol=2e-3, atol=2e-4 * float(np.abs(g[f'{tag}_grad.{k}']).max() + 1e-6), err_msg=k)
            else:
                st = g[f'{tag}_gradstat.{k}']
                gg = p.grad.double().flatten()
                np.testing.assert_allclose(float(gg.norm()), st[1], rtol=1e-3, err_msg=k)
                np.testing.assert_allclose(gg[:6].numpy(), st[2:], rtol=5e-3, atol=1e-3 * st[1], err_msg=k)


def test_objectives(golden):
    g = golden('objectives')
    a = stub_cqt.closed_form_coefficients(2, 540, 5).requires_grad_(True)
    b = (stub_cqt.closed_form_coefficients(2, 540, 5) * 0.7 + 0.1).flip(-1).requires_grad_(True)
    l = oobj.compute_reconstruction_loss(a, b)
    ga, gb = torch.autograd.grad(l, (a, b))
    np.testing.assert_allclose(float(l), float(g['rec_loss']), rtol=1e-6)
    np.testing.assert_allclose(ga[:, :, ::45].numpy(), g['rec_ga'], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(gb[:, :, ::45].numpy(), g['rec_gb'], rtol=1e-5, atol=1e-7)
    est = torch.sigmoid(stub_cqt.closed_form_coefficients(2, 540, 7)[:, 0]).requires_grad_(True)
    tgt = stub_cqt.closed_form_targets(2, 540, 7)
    for w in (False, True):
        l = oobj.compute_transcription_loss(est, tgt, w)
        gr, = torch.autograd.grad(l, est)
        np.testing.assert_allclose(float(l), float(g[f'trn_loss_w{int(w)}']), rtol=1e-6)
        np.testing.assert_allclose(gr.numpy(), g[f'trn_grad_w{int(w)}'], rtol=1e-5, atol=1e-8)
    l = oobj.compute_transcription_loss(est[:1, :, :2], torch.ones(1, 540, 2), True)
    np.testing.assert_allclose(float(l), float(g['trn_loss_allones']), rtol=1e-6)
    sp, sc = oobj.compute_consistency_loss(a, b, (a + b) / 2)
    np.testing.assert_allclose([float(sp), float(sc)], g['cons'], rtol=1e-6)


def test_wrapper_arithmetic_bit_exact(golden):
    g = golden('wrapper')
    N, M, sr = 66150, 1024, 22050
    assert float(g['hop_length']) == N / M == 64.599609375
    assert np.array_equal(nsgt.get_midi_freqs(9, 60, sr), g['midi_freqs'])
    assert [nsgt.get_expected_frames(int(n), N, M) for n in g['frames_in']] == list(g['frames_out'])
    assert [nsgt.get_expected_samples(float(t), sr) for t in g['samples_in']] == list(g['samples_out'])
    assert np.array_equal(nsgt.get_times(3100, N, M, sr), g['times_3100'])
    assert [nsgt.pad_to_block_length(np.zeros((1, 1, int(n))), N).shape[-1] for n in g['pad_lens_in']] == list(g['pad_lens_out'])
    c = (g['to_real_in_re'] + 1j * g['to_real_in_im'])
    r = nsgt.to_real(c)
    assert np.array_equal(r, g['to_real_out'])
    np.testing.assert_allclose(nsgt.to_magnitude(r), g['to_magnitude_out'], rtol=1e-6)
    cc = nsgt.to_complex(r)
    assert np.array_equal(cc.real, g['to_complex_re']) and np.array_equal(cc.imag, g['to_complex_im'])
