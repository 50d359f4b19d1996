"""Diagnostic (round 4): gradient of decoder.block4.block1.conv1.0.bias at the reference's training shape (3 items x 3 blocks, 2 annotated)
on the fp32 HIP path and under autocast; run with different TTRAP_* switches to separate kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'timbre-trap_amd'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    sys.path.insert(0, p)
import torch
from timbre_trap.framework import TimbreTrap, compute_consistency_loss, compute_reconstruction_loss, compute_transcription_loss
import test_gpu_model as tg

n_clips, n_blocks, n_mpe = 3, int(os.environ.get('DIAG_BLOCKS', '3')), 2
T = n_blocks * 1024
torch.manual_seed(2)
model = TimbreTrap(22050, 9, 60, 3, latent_size=128, model_complexity=2).cuda()
g_audio = torch.Generator().manual_seed(1234)
audio = torch.rand(n_clips, 1, n_blocks * 66150, generator=g_audio) * 2 - 1
with torch.no_grad():
    c = model.sliCQ(audio.cuda())
g = tg._bench_style_targets(n_mpe, 540, T).cuda()
names = [k for k, _ in model.named_parameters() if k.endswith('bias') and ('decoder.block4' in k or 'encoder.block1' in k)]
res = {}
for amp in (False, True):
    with torch.autocast(device_type='cuda', dtype=torch.bfloat16, enabled=amp):
        latents, emb, _ = model.encoder(c)
        rec, trn = model.decode(latents, None), model.decode(latents, None, True)
        lat2, _, _ = model.encoder(trn)
        trn_rec, trn_scr = model.decode(lat2, None), model.decode(lat2, None, True)
        act = model.to_activations(trn)
        total = compute_reconstruction_loss(rec, c) + compute_transcription_loss(act[:n_mpe], g, True) + sum(compute_consistency_loss(trn_rec[:n_mpe], trn_scr[:n_mpe], trn[:n_mpe]))
        model.zero_grad()
        total.backward()
    res[amp] = {k: p.grad.detach().double().cpu().clone() for k, p in model.named_parameters()}
for k in names:
    a, b = res[False][k], res[True][k]
    print('%-40s fp32 %s | bf16 %s | rel %.3e' % (k, ' '.join('%+.4e' % v for v in a.flatten()[:8]), ' '.join('%+.4e' % v for v in b.flatten()[:8]),
                                                 float((a - b).norm() / a.norm())))
worst = sorted(((float((res[False][k] - res[True][k]).norm() / (res[False][k].norm() + 1e-30)), k) for k in res[False]), reverse=True)[:6]
print('worst vs fp32 HIP path:', worst)
