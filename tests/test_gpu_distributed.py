"""
N > 1 train-step path on the GPU box: two ranks (gloo backend, so that both can share the single GPU of the test box; the
real launch uses RCCL, one GPU per rank) run FusedAdamW steps with ONE all-reduce of the flat gradient and must end with
identical parameters, equal to a single process that sees the concatenated batch (mean of rank means == global mean).
"""

import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'timbre-trap_amd'))
    import torch
    from timbre_trap.framework import TimbreTrap, compute_reconstruction_loss
    from timbre_trap.utils import FusedAdamW, init_process_group_from_env, allreduce_gradients
    from timbre_trap.utils.distributed import broadcast_parameters
    rank, world, _ = init_process_group_from_env()
    torch.cuda.set_device(0)
    torch.manual_seed(7 + rank)                      # different initial weights per rank: the broadcast must fix that
    model = TimbreTrap(sample_rate=22050, n_octaves=9, bins_per_octave=60, latent_size=32, model_complexity=1).cuda()
    opt = FusedAdamW(model.parameters(), lr=1e-3, max_norm=10.0)
    broadcast_parameters(opt.flat_param)
    g = torch.Generator().manual_seed(123)
    audio_all = torch.rand(2, 1, 66150, generator=g) * 2 - 1
    audio = audio_all[rank:rank + 1].cuda() if world > 1 else audio_all.cuda()
    for _ in range(2):
        coeffs = model.sliCQ(audio)
        rec = model(audio, False)[0]
        loss = compute_reconstruction_loss(rec, coeffs)
        opt.zero_grad()
        loss.backward()
        if world > 1:
            allreduce_gradients(opt.flat_grad, world)
        opt.step()
    print('RESULT %%d %%.9e %%.9e' %% (rank, float(opt.flat_param.double().sum()), float(opt.flat_param.double().abs().sum())))
''') % (ROOT, ROOT)


def _run(rank, world, port):
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
               TTRAP_DIST_BACKEND='gloo')
    return subprocess.Popen([sys.executable, '-c', SCRIPT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


def _result(proc):
    out, _ = proc.communicate(timeout=600)
    assert proc.returncode == 0, out[-2000:]
    line = [l for l in out.splitlines() if l.startswith('RESULT')][-1].split()
    return float(line[2]), float(line[3])


@pytest.mark.gpu
def test_two_ranks_match_single_process_global_batch():
    procs = [_run(r, 2, 29541) for r in range(2)]
    res = [_result(p) for p in procs]
    assert res[0] == res[1]                                       # ranks end with bit-identical parameters
    single = _result(_run(0, 1, 29542))
    assert abs(res[0][0] - single[0]) <= 1e-5 * abs(single[1]) and abs(res[0][1] - single[1]) <= 1e-5 * abs(single[1])
