"""
GPU parity of the whole path: autoencoder (golden fixtures recorded from the reference + the CPU oracle),
objectives, the train step (losses -> backward -> fused clip + AdamW) and chunked inference.
"""

import numpy as np
import pytest
import torch

import stub_cqt
from oracle import autoencoder as oae
from oracle import nsgt
from oracle import objectives as oobj
from oracle.train_step import OracleTrainer

pytestmark = pytest.mark.gpu
N, M, SR = 66150, 1024, 22050
LOGIT_TOL = dict(rtol=1e-4, atol=1e-4)


def _model(tag_kw, sd=None):
    from timbre_trap.framework import TimbreTrap
    m = TimbreTrap(SR, 9, 60, 3, **tag_kw)
    if sd is not None:
        m.load_state_dict(sd, strict=True)
    return m.cuda()


KW = dict(mc1=dict(latent_size=None, model_complexity=1, skip_connections=False),
          mc2skip=dict(latent_size=128, model_complexity=2, skip_connections=True))


def test_encoder_decoder_golden(golden):
    g = golden('encdec')
    from timbre_trap.framework import Decoder, Encoder
    for mc, lat in ((1, None), (2, 128)):
        enc, dec = Encoder(540, lat, mc), Decoder(540, lat, mc)
        shapes = oae.state_dict_shapes(540, lat, mc)
        enc.load_state_dict(oae.closed_form_state_dict({k[8:]: v for k, v in shapes.items() if k.startswith('encoder.')}))
        dec.load_state_dict(oae.closed_form_state_dict({k[8:]: v for k, v in shapes.items() if k.startswith('decoder.')}))
        enc, dec = enc.cuda(), dec.cuda()
        coeffs = stub_cqt.closed_form_coefficients(1, 540, 6).cuda()
        latents, emb, losses = enc(coeffs)
        assert losses == {}
        np.testing.assert_allclose(latents.detach().cpu().numpy(), g[f'mc{mc}_latents'], **LOGIT_TOL)
        for i, e in enumerate(emb):
            np.testing.assert_allclose(e.detach().cpu().numpy(), g[f'mc{mc}_emb{i}'], **LOGIT_TOL)
        ind = torch.ones_like(latents[..., :1, :])
        np.testing.assert_allclose(dec(torch.cat((latents, ind), -2)).detach().cpu().numpy(), g[f'mc{mc}_dec'], **LOGIT_TOL)
        np.testing.assert_allclose(dec(torch.cat((latents, 0 * ind), -2), emb).detach().cpu().numpy(), g[f'mc{mc}_dec_skip'], **LOGIT_TOL)


@pytest.mark.parametrize('tag', ['mc1', 'mc2skip'])
def test_model_forward_losses_gradients_golden(golden, tag):
    """Autoencoder + objectives + gradients vs values recorded from the reference (coefficients from the stub)."""
    from timbre_trap.framework import compute_consistency_loss, compute_reconstruction_loss, compute_transcription_loss
    g = golden('model')
    sd = oae.closed_form_state_dict(oae.state_dict_shapes(540, **KW[tag]))
    model = _model(KW[tag], sd)
    audio = stub_cqt.closed_form_audio(2, 64)
    coeffs = torch.from_numpy(nsgt.to_real(stub_cqt.stub_encode(audio, 540, 64, 16).numpy())).contiguous().cuda()
    # TimbreTrap.forward with the CQT replaced by the recorded stub coefficients
    latents, emb, _ = model.encoder(coeffs)
    emb = model.apply_skip_connections(emb)
    rec, trn = model.decode(latents, emb), model.decode(latents, emb, True)
    lat2, emb2, _ = model.encoder(trn)
    emb2 = model.apply_skip_connections(emb2)
    trn_rec, trn_scr = model.decode(lat2, emb2), model.decode(lat2, emb2, True)
    for name, t in (('reconstruction', rec), ('latents', latents), ('transcription', trn),
                    ('transcription_rec', trn_rec), ('transcription_scr', trn_scr)):
        np.testing.assert_allclose(t.detach().cpu().numpy(), g[f'{tag}_fwd_{name}'], err_msg=name, **LOGIT_TOL)
    act = model.to_activations(trn)
    np.testing.assert_allclose(act.detach().cpu().numpy(), g[f'{tag}_act'], rtol=1e-4, atol=1e-5)
    gt = stub_cqt.closed_form_targets(2, 540, 16).cuda()
    l_rec = compute_reconstruction_loss(rec, coeffs)
    l_trn = compute_transcription_loss(act, gt, True)
    l_sp, l_sc = compute_consistency_loss(trn_rec, trn_scr, trn)
    total = l_rec + l_trn + (l_sp + l_sc)
    got = [float(v) for v in (l_rec, l_trn, l_sp, l_sc, total)]
    np.testing.assert_allclose(got, g[f'{tag}_losses'], rtol=1e-4)
    model.zero_grad()
    total.backward()
    for k, p in model.named_parameters():
        if tag == 'mc1':
            ref = g[f'{tag}_grad.{k}']
            np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=5e-3, atol=5e-4 * float(np.abs(ref).max() + 1e-6), err_msg=k)
        else:
            st = g[f'{tag}_gradstat.{k}']
            gg = p.grad.double().flatten().cpu()
            np.testing.assert_allclose(float(gg.norm()), st[1], rtol=2e-3, err_msg=k)
            np.testing.assert_allclose(gg[:6].numpy(), st[2:], rtol=1e-2, atol=2e-3 * st[1], err_msg=k)


def test_objectives_golden(golden):
    from timbre_trap.framework import compute_consistency_loss, compute_reconstruction_loss, compute_transcription_loss
    g = golden('objectives')
    a = stub_cqt.closed_form_coefficients(2, 540, 5).cuda().requires_grad_(True)
    b = (stub_cqt.closed_form_coefficients(2, 540, 5) * 0.7 + 0.1).flip(-1).contiguous().cuda().requires_grad_(True)
    l = compute_reconstruction_loss(a, b)
    ga, gb = torch.autograd.grad(l, (a, b))
    np.testing.assert_allclose(float(l), float(g['rec_loss']), rtol=1e-5)
    np.testing.assert_allclose(ga[:, :, ::45].cpu().numpy(), g['rec_ga'], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(gb[:, :, ::45].cpu().numpy(), g['rec_gb'], rtol=1e-5, atol=1e-7)
    est = torch.sigmoid(stub_cqt.closed_form_coefficients(2, 540, 7)[:, 0]).cuda().requires_grad_(True)
    tgt = stub_cqt.closed_form_targets(2, 540, 7).cuda()
    for w in (False, True):
        l = compute_transcription_loss(est, tgt, w)
        gr, = torch.autograd.grad(l, est)
        np.testing.assert_allclose(float(l), float(g[f'trn_loss_w{int(w)}']), rtol=1e-5)
        np.testing.assert_allclose(gr.cpu().numpy(), g[f'trn_grad_w{int(w)}'], rtol=1e-5, atol=1e-8)
    l = compute_transcription_loss(est[:1, :, :2].contiguous(), torch.ones(1, 540, 2, device='cuda'), True)
    np.testing.assert_allclose(float(l), float(g['trn_loss_allones']), rtol=1e-5)
    sp, sc = compute_consistency_loss(a, b, ((a + b) / 2).detach())
    np.testing.assert_allclose([float(sp), float(sc)], g['cons'], rtol=1e-5)


def test_full_path_vs_oracle_real_cqt():
    """audio -> HIP CQT -> HIP autoencoder (consistency) at T=1024 vs the oracle end to end (mc=1, B=1)."""
    tab = nsgt.nsgt_tables(9, 60, SR, N)
    sd = oae.closed_form_state_dict(oae.state_dict_shapes(540, **KW['mc1']))
    model = _model(KW['mc1'], sd)
    g = torch.Generator().manual_seed(1234)
    audio = torch.rand(1, 1, N, generator=g) * 2 - 1
    audio = audio / audio.abs().max()
    res = model(audio.cuda(), consistency=True)
    assert res[5] == {} and len(res) == 6
    coeffs = torch.from_numpy(np.ascontiguousarray(nsgt.wrapper_forward(audio.numpy(), tab))).float()
    ref = oae.forward(coeffs, sd, consistency=True)
    for name, got, want in zip(('reconstruction', 'latents', 'transcription', 'transcription_rec', 'transcription_scr'), res, ref):
        err = float((got.cpu() - want).abs().max() / want.abs().max())
        assert err < 1e-4, (name, err)
    nc = model(audio.cuda(), consistency=False)
    assert nc[3] is None and nc[4] is None and torch.equal(nc[0], res[0])


def test_chunked_inference_transcribe_reconstruct():
    tab = nsgt.nsgt_tables(9, 60, SR, N)
    sd = oae.closed_form_state_dict(oae.state_dict_shapes(540, **KW['mc1']))
    model = _model(KW['mc1'], sd).eval()
    g = torch.Generator().manual_seed(5)
    audio = torch.rand(1, 1, int(1.5 * N), generator=g) * 2 - 1

    def cqt_fwd(chunk):
        return torch.from_numpy(np.ascontiguousarray(nsgt.wrapper_forward(chunk.numpy(), tab))).float()
    for transcribe in (True, False):
        want = oae.chunked_inference(audio, sd, cqt_fwd, N, M, transcribe)
        got = model.chunked_inference(audio.cuda(), transcribe)
        assert got.shape == want.shape == (1, 2, 540, 2 * M)                 # bit-exact frame count
        assert float((got.cpu() - want).abs().max() / want.abs().max()) < 1e-4
    act = model.transcribe(audio.cuda())
    assert act.shape == (1, 540, 2 * M) and float(act.min()) >= 0 and float(act.max()) < 1
    rec = model.reconstruct(audio.cuda())
    assert rec.shape == (1, 1, 2 * N) and abs(float(rec.abs().max()) - 1.0) < 1e-5
    # reconstruct == decode(chunked_inference), bit for bit.  (Decode accuracy itself is pinned on well-conditioned
    # inputs in tests/test_gpu_cqt.py: a random-weight network emits time-smooth coefficients that the synthesis
    # windows almost entirely reject, so what is left sits at the fp32 round-off level of the 1024-point FFTs and
    # cannot be compared with a float64 oracle.)
    coeffs = model.chunked_inference(audio.cuda(), False)
    assert torch.equal(model.sliCQ.decode(coeffs), rec)
    full = model.inference(audio.cuda(), True)
    assert full.shape == (1, 2, 540, 2 * M)


def test_train_steps_match_oracle():
    """Two optimisation steps (clip 10 + AdamW) on the HIP path track the CPU oracle trainer."""
    from timbre_trap.framework import compute_consistency_loss, compute_reconstruction_loss, compute_transcription_loss
    from timbre_trap.utils import FusedAdamW
    kw = KW['mc1']
    sd = oae.closed_form_state_dict(oae.state_dict_shapes(540, **kw), amplitude=0.12)
    model = _model(kw, sd)
    opt = FusedAdamW(model.parameters(), lr=1e-3, max_norm=10.0)
    oracle = OracleTrainer(sd, lr=1e-3)
    T = 64
    for step in range(2):
        coeffs = stub_cqt.closed_form_coefficients(2, 540, T) * (1.0 + 0.1 * step)
        gt = stub_cqt.closed_form_targets(2, 540, T)
        ref = oracle.step(coeffs, gt)
        c = coeffs.cuda()
        latents, emb, _ = model.encoder(c)
        rec, trn = model.decode(latents, None), model.decode(latents, None, True)
        lat2, _, _ = model.encoder(trn)
        trn_rec, trn_scr = model.decode(lat2, None), model.decode(lat2, None, True)
        act = model.to_activations(trn)
        l_rec = compute_reconstruction_loss(rec, c)
        l_trn = compute_transcription_loss(act, gt.cuda(), True)
        l_sp, l_sc = compute_consistency_loss(trn_rec, trn_scr, trn)
        total = l_rec + l_trn + (l_sp + l_sc)
        opt.zero_grad()
        total.backward()
        norm = opt.step()
        np.testing.assert_allclose(float(total), ref['total'], rtol=2e-4)
        np.testing.assert_allclose(float(norm), ref['grad_norm'], rtol=2e-3)
    for k, p in model.named_parameters():
        want = oracle.params[k].detach()
        assert float((p.detach().cpu() - want).abs().max()) < 2e-4, k
