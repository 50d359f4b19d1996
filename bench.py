"""
bench.py -- audio-seconds/s training throughput of the Timbre-Trap hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by the driver as  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One step = exactly the body of reference experiments/train.py:404-496 on one batch of synthetic
audio already resident in HBM:
    coefficients = model.sliCQ(audio)                         (CQT forward, target of the reconstruction loss)
    model(audio, consistency=True)                            (CQT again inside encode, 2 encoder + 4 decoder passes)
    to_activations, reconstruction / transcription / 2 consistency losses, total
    zero_grad, backward, [all-reduce of the flat gradient when N > 1], clip_grad_norm_(10) + AdamW
Workload (BASELINE.json configs[2]): model_complexity=2, latent_size=128, 64 clips x 3 s @ 22.05 kHz per
GPU, 9 octaves x 60 bins/octave; weak scaling (per-GPU batch fixed).  dtype = fp32 (exact-fp32 MFMA path).

Prints ONE JSON line on rank 0 with the driver's fields plus
  "roofline"     : dominant kernel, algorithmic FLOPs / average launch time measured with HIP events on the
                   launch stream over the timed steps, against the gfx950 fp32 matrix peak
  "cpu_baseline" : the CPU oracle (kind "port") timed on this box's host cores on a bounded sample (rank 0, N = 1)
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'timbre-trap_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

N_BLOCK, M_FRAMES, SR, N_BINS = 66150, 1024, 22050, 540
SECS_PER_CLIP = 3.0
PEAK_FP32_MATRIX_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 peak
PEAK_HBM_GBS = 8000.0


def synthetic_batch(batch, rank, device='cpu'):
    """
    SURVEY.md section 8d: audio ~ U(-1,1) (seed 1234+rank) inf-norm normalised per item
    (mirrors reference AudioDataset.py:75-77); targets = Bernoulli(0.01) seeds (seed 4321+rank) blurred along
    frequency with a sigma = 1 bin Gaussian, re-normalised so seeds are exactly 1.0 and clipped to [0,1]
    (mirrors reference PitchDataset.py:297-305).
    """
    g = torch.Generator().manual_seed(1234 + rank)
    audio = torch.rand(batch, 1, N_BLOCK, generator=g) * 2 - 1
    audio = audio / audio.abs().amax(dim=-1, keepdim=True)
    g2 = torch.Generator().manual_seed(4321 + rank)
    seeds = (torch.rand(batch, N_BINS, M_FRAMES, generator=g2) < 0.01).float()
    k = torch.exp(-0.5 * torch.arange(-4, 5, dtype=torch.float32) ** 2).view(1, 1, 9)
    blurred = torch.nn.functional.conv1d(seeds.permute(0, 2, 1).reshape(-1, 1, N_BINS), k, padding=4)
    blurred = blurred.reshape(batch, M_FRAMES, N_BINS).permute(0, 2, 1)
    target = torch.maximum(blurred.clamp(0, 1), seeds).contiguous()
    return audio.to(device), target.to(device)


def build_model(mc, latent, device, seed=2):
    from timbre_trap.framework import TimbreTrap
    torch.manual_seed(seed)                       # reference experiments/train.py:88,137
    return TimbreTrap(sample_rate=SR, n_octaves=9, bins_per_octave=60, secs_per_block=3,
                      latent_size=latent, model_complexity=mc, skip_connections=False).to(device)


def train_step(model, opt, audio, target, world):
    from timbre_trap.framework import compute_consistency_loss, compute_reconstruction_loss, compute_transcription_loss
    from timbre_trap.utils import allreduce_gradients
    coefficients = model.sliCQ(audio)
    reconstruction, latents, trn_coeffs, trn_rec, trn_scr, _ = model(audio, True)
    transcription = model.to_activations(trn_coeffs)
    n = target.size(0)
    l_rec = compute_reconstruction_loss(reconstruction, coefficients)
    l_trn = compute_transcription_loss(transcription[:n], target, True)
    l_sp, l_sc = compute_consistency_loss(trn_rec[:n], trn_scr[:n], trn_coeffs[:n])
    total = l_rec + l_trn + (l_sp + l_sc)
    opt.zero_grad()
    total.backward()
    if world > 1:
        allreduce_gradients(opt.flat_grad, world)
    opt.step()
    return total


def available_cores():
    """Cores this process may actually use: scheduler affinity capped by the cgroup CPU quota (containers)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    period = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                    n = min(n, max(1, q // period))
        except (OSError, ValueError, IndexError):
            pass
    return n


def cpu_baseline(mc, latent, seconds_budget=20.0):
    """The oracle (CPU restatement, torch + numpy on the host cores) on a bounded sample of the same workload."""
    from oracle import nsgt
    from oracle.train_step import OracleTrainer, cqt_forward_torch
    threads = min(available_cores(), 64)          # beyond ~64 threads the small conv layers only lose to sync overhead
    torch.set_num_threads(threads)
    model = build_model(mc, latent, 'cpu')
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    trainer = OracleTrainer(sd, lr=1e-3)
    tab = nsgt.nsgt_tables(9, 60, SR, N_BLOCK)
    audio, target = synthetic_batch(1, 0)

    def one():
        coeffs = cqt_forward_torch(audio, tab)           # train.py:404
        _ = cqt_forward_torch(audio, tab)                # the transform again inside model.encode (modules.py:88)
        trainer.step(coeffs, target)
    t0 = time.perf_counter()
    one()                                                # warm-up (also the sample if the host is very slow)
    warm = time.perf_counter() - t0
    n, el = 0, 0.0
    t0 = time.perf_counter()
    while warm < seconds_budget and el + warm < seconds_budget and n < 5:
        one()
        n += 1
        el = time.perf_counter() - t0
    step_s = el / n if n else warm
    n = max(n, 1)
    cpu_model = ''
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                cpu_model = line.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    return dict(value=SECS_PER_CLIP / step_s, unit='audio-seconds/s', cores=threads, kind='port',
                sample='%d train steps of 1 clip x 3 s (mc=%d, latent=%s) after 1 warm-up, fp32, %d torch threads, %s' % (n, mc, latent, threads, cpu_model),
                s_per_step=step_s)


def bench_inference(model, args, rank, world, dev):
    """BASELINE.json configs[1]: model.transcribe(audio) + model.reconstruct(audio), batch x 3 s clips (secondary line)."""
    model.eval()
    batch = 32 if args.batch == 64 else args.batch
    audio, _ = synthetic_batch(batch, rank, dev)

    def step():
        with torch.no_grad():
            act = model.transcribe(audio)
            rec = model.reconstruct(audio)
        return act, rec
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        act, rec = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms = 1000.0 * elapsed / args.steps
    if rank == 0:
        print(json.dumps(dict(metric='audio-seconds/s inference throughput, transcribe()+reconstruct() (9oct x 60bpo, 3s@22.05kHz)',
                              value=world * batch * SECS_PER_CLIP / (elapsed / args.steps), unit='audio-seconds/s', n_gpus=world,
                              steps=args.steps, warmup=args.warmup, ms_per_step=ms, higher_is_better=True, scaling='weak',
                              vs_baseline=None, dtype={'fp32': 'f32', 'bf16x3': 'bf16x3', 'bf16': 'bf16'}[args.precision], data='synthetic',
                              config=dict(workload='transcribe() + reconstruct() (each: 3 half-overlapping chunks per clip through CQT + '
                                                   'encoder + decoder, Hann cross-fade; reconstruct adds the inverse CQT), model_complexity=%d '
                                                   'latent=%d, %d clips x 3 s' % (args.mc, args.latent, batch),
                                          global_batch=world * batch, parallelism='dp%d' % world),
                              out_shapes=[list(act.shape), list(rec.shape)])))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=64, help='clips per GPU')
    ap.add_argument('--mc', type=int, default=2)
    ap.add_argument('--latent', type=int, default=128)
    ap.add_argument('--precision', choices=('fp32', 'bf16x3', 'bf16'), default=os.environ.get('TTRAP_PRECISION', 'fp32'),
                    help='operands of the wide 3x3 convs on the matrix cores: fp32 = exact (default), bf16 = rounded operands, fp32 accumulation')
    ap.add_argument('--mode', choices=('train', 'infer'), default='train',
                    help="train = the headline metric; infer = BASELINE config[1]: transcribe() + reconstruct() on 32 clips x 3 s")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    from timbre_trap import _hip
    from timbre_trap.framework import ops
    from timbre_trap.utils import FusedAdamW, init_process_group_from_env
    ops.PRECISION = args.precision
    from timbre_trap.utils.distributed import broadcast_parameters
    import torch.distributed as dist

    rank, world, local_rank = init_process_group_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (there is no CPU fallback for the HIP path)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if not os.path.exists(_hip.LIB_PATH):       # normally prebuilt by __graft_entry__.build(); compile once per node otherwise
        if local_rank == 0:
            _hip.build()
        if world > 1:
            dist.barrier()
    _hip.lib()

    model = build_model(args.mc, args.latent, dev)
    if args.mode == 'infer':
        return bench_inference(model, args, rank, world, dev)
    opt = FusedAdamW(model.parameters(), lr=1e-3, max_norm=10.0)
    if world > 1:
        broadcast_parameters(opt.flat_param)
    audio, target = synthetic_batch(args.batch, rank, dev)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        train_step(model, opt, audio, target, world)
    sync()
    _hip.EVENT_LOG = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        total = train_step(model, opt, audio, target, world)
    sync()
    elapsed = time.perf_counter() - t0
    events = _hip.EVENT_LOG
    _hip.EVENT_LOG = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms = 1000.0 * elapsed / args.steps
    value = world * args.batch * SECS_PER_CLIP / (elapsed / args.steps)

    if rank == 0:
        # dominant kernel: the fused residual block at the widest level (C = 16*mc channels, H = 65 rows)
        C = 16 * 2 ** (args.mc - 1)
        key = 'resblock_fwd_C%d' % C
        roof = None
        if events.get(key):
            times = [s.elapsed_time(e) for s, e in events[key]]
            avg_ms = sum(times) / len(times)
            flops = 2.0 * (9 * C * C + C * C) * args.batch * 65 * M_FRAMES
            ach = flops / (avg_ms * 1e-3) / 1e12
            # HBM traffic per launch of this kernel from the rocprofv3 PMC passes committed under profiles/
            # (FETCH_SIZE x2 per the gfx950 guide + WRITE_SIZE); only valid for the shape it was measured on
            traffic = None
            pmc = os.path.join(ROOT, 'profiles', 'r01_f_pmc_rb_fwd_C32.json')
            if C == 32 and args.batch == 64 and args.precision == 'fp32' and os.path.exists(pmc):
                traffic = json.load(open(pmc))['traffic_bytes_corrected']
            roof = dict(kernel='k_rb_fwd<%d,D> (fused ResidualConv2dBlock forward, C=%d, H=65; same MFMA main loop as the '
                               'data-gradient kernel k_conv_mfma)' % (C, C),
                        bound='mfma', achieved=ach, peak=PEAK_FP32_MATRIX_TFLOPS, unit='TFLOP/s',
                        frac=ach / PEAK_FP32_MATRIX_TFLOPS, traffic=traffic, algorithmic_flops=flops,
                        algorithmic_bytes=2.0 * 4 * C * args.batch * 65 * M_FRAMES, launches=len(times), avg_ms=avg_ms)
        cqt = None
        if events.get('cqt_forward'):
            times = [s.elapsed_time(e) for s, e in events['cqt_forward']]
            avg_ms = sum(times) / len(times)
            gbs = args.batch * 4688280 / (avg_ms * 1e-3) / 1e9
            cqt = dict(kernel='tt_cqt_forward (3 launches)', bound='hbm', achieved=gbs, peak=PEAK_HBM_GBS, unit='GB/s',
                       frac=gbs / PEAK_HBM_GBS, avg_ms=avg_ms)
        base = None
        if not args.no_cpu_baseline and world == 1:
            base = cpu_baseline(args.mc, args.latent)
        line = dict(metric='audio-seconds/s training throughput (9oct x 60bpo, 3s@22.05kHz)', value=value,
                    unit='audio-seconds/s', n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=ms,
                    higher_is_better=True, scaling='weak', vs_baseline=None, dtype={'fp32': 'f32', 'bf16x3': 'bf16x3', 'bf16': 'bf16'}[args.precision], data='synthetic',
                    config=dict(workload='full train step (CQT x2 + AE fwd/bwd with consistency + 3 losses + clip + AdamW), '
                                         'model_complexity=%d latent=%d, %d clips x 3 s per GPU' % (args.mc, args.latent, args.batch),
                                global_batch=world * args.batch, parallelism='dp%d' % world),
                    roofline=roof, roofline_cqt=cqt, cpu_baseline=base, final_loss=float(total))
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
