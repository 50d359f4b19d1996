"""Round 4, verdict item 8 -- what fp16 (the reference's own autocast dtype, train.py:415) buys over bf16 on this network: the train step of
mc 2 / latent 128 with consistency, default init, on the fp32 HIP path (within 1e-4 of the CPU restatement) and under autocast in both
16-bit element types; agreement of the five outputs, the losses and all 120 parameter gradients with fp32, for DIAG_B clips (default 2 and 64:
the loss is a mean over B x T frames, so dL/dlogit ~ 2 err / (B T) -- 3e-5 err at 64 clips, inside fp16's subnormal range, no GradScaler in the
reference)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'timbre-trap_amd'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    sys.path.insert(0, p)
import torch
from timbre_trap.framework import TimbreTrap, compute_consistency_loss, compute_reconstruction_loss, compute_transcription_loss
import bench

for B in [int(v) for v in os.environ.get('DIAG_B', '2,64').split(',')]:
    torch.manual_seed(2)
    model = TimbreTrap(22050, 9, 60, 3, latent_size=128, model_complexity=2).cuda()
    audio, target = bench.synthetic_batch(B, 0, 'cuda')
    with torch.no_grad():
        c = model.sliCQ(audio)
    res = {}
    from timbre_trap.framework import ops
    scales = [float(v) for v in os.environ.get('DIAG_SCALES', '1,4096,65536').split(',')]
    runs = [('fp32', None, 1.0), ('bf16', torch.bfloat16, 1.0)] + [('fp16 S=%g' % sc, torch.float16, sc) for sc in scales]
    for name, dt, sc in runs:
        ops.FP16_LOSS_SCALE = sc                    # static loss scale of the fp16 backward (round 5; 1 = the reference's own arithmetic)
        with torch.autocast(device_type='cuda', dtype=dt or torch.bfloat16, enabled=dt is not None):
            latents, emb, _ = model.encoder(c)
            rec, trn = model.decode(latents, None), model.decode(latents, None, True)
            lat2, _, _ = model.encoder(trn)
            trn_rec, trn_scr = model.decode(lat2, None), model.decode(lat2, None, True)
            act = model.to_activations(trn)
            losses = [compute_reconstruction_loss(rec, c), compute_transcription_loss(act, target, True), *compute_consistency_loss(trn_rec, trn_scr, trn)]
            total = losses[0] + losses[1] + (losses[2] + losses[3])
            model.zero_grad()
            total.backward()
        torch.cuda.synchronize()
        res[name] = dict(outs=[t.detach().float().clone() for t in (rec, latents, trn, trn_rec, trn_scr)], losses=[float(l) for l in losses],
                         grads={k: p.grad.detach().double().clone() for k, p in model.named_parameters()},
                         amax=[float(t.detach().float().abs().max()) for t in (rec, latents, trn)])
        del latents, emb, rec, trn, lat2, trn_rec, trn_scr, act, total
    ref = res['fp32']
    print('== %d clips x 3 s (T = 1024), mc 2 / latent 128, default init; reference = fp32 HIP path; |logits| max %.1f, |latents| max %.1f' % (B, ref['amax'][0], ref['amax'][1]))
    for name in [n for n, _, _ in runs[1:]]:
        r = res[name]
        out_err = [float((a - b).abs().max() / b.abs().max()) for a, b in zip(r['outs'], ref['outs'])]
        loss_err = [abs(a - b) / abs(b) for a, b in zip(r['losses'], ref['losses'])]
        rels = sorted((float((r['grads'][k] - ref['grads'][k]).norm() / (ref['grads'][k].norm() + 1e-300)), k) for k in ref['grads'])
        nonfinite = [k for k in r['grads'] if not bool(torch.isfinite(r['grads'][k]).all())]
        zeros = [k for k in r['grads'] if float(r['grads'][k].abs().max()) == 0.0]
        print('  %s: outputs rel max %s | losses rel %s' % (name, ' '.join('%.2e' % e for e in out_err), ' '.join('%.2e' % e for e in loss_err)))
        print('        gradients rel L2: median %.3e  90%% %.3e  worst %.3e (%s); non-finite %d, all-zero %d' % (
            rels[len(rels) // 2][0], rels[int(0.9 * len(rels))][0], rels[-1][0], rels[-1][1], len(nonfinite), len(zeros)))
