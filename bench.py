"""
bench.py -- audio-seconds/s training throughput of the Timbre-Trap hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    N > 1: one process per GPU.  Either the caller starts the ranks (python -m torch.distributed.run --nproc-per-node N ...
    bench.py --gpus N ..., WORLD_SIZE in the environment), or the bare command above starts them itself: before anything
    touches the GPU it builds the library and runs torch.distributed.run as a CHILD process (never an exec), relays rank 0's
    JSON line and exits with the child's code.  It never falls back to fewer GPUs: --gpus N with fewer than N visible
    devices exits non-zero (TTRAP_DIST_BACKEND=gloo lets the ranks share the visible device(s): functional test only).

One step = exactly the body of reference experiments/train.py:404-496 on one batch of synthetic
audio already resident in HBM:
    coefficients = model.sliCQ(audio)                         (CQT forward, target of the reconstruction loss)
    model(audio, consistency=True)                            (CQT again inside encode, 2 encoder + 4 decoder passes)
    to_activations, reconstruction / transcription / 2 consistency losses, total
    zero_grad, backward, [all-reduce of the flat gradient when N > 1], clip_grad_norm_(10) + AdamW
Workload (BASELINE.json configs[2]): model_complexity=2, latent_size=128, 64 clips x 3 s @ 22.05 kHz per
GPU, 9 octaves x 60 bins/octave; weak scaling (per-GPU batch fixed).  The step runs under torch.autocast like the reference's
(experiments/train.py:415): dtype = bf16 (bf16 channels-last MFMA conv path, fp32 master weights / losses / optimizer);
`--precision fp32` times the exact-fp32 path that carries the 1e-4 output bar (also reported as "fp32_train_step").

Prints ONE JSON line on rank 0 with the driver's fields plus
  "roofline"          : the by-time dominant call of the step -- the residual-block backward at the widest level
                        (tt_wide_rb_bwd, C = 32: k_wrb_bwd_a + k_wrb_dxw + k_wrb_reduce, 18 calls per step):
                        algorithmic bytes (dy, x read, dx written once, bf16) / average call time measured with HIP events on the
                        launch stream over the timed steps, against the HBM peak; `traffic` is the HBM byte count of the committed
                        rocprofv3 PMC passes of those kernels (`traffic_source` names the file -- not re-measured in this run)
  "roofline_fwd"      : the fused block forward k_wrb_conv<32,D,0> the same way (round 2's roofline kernel; fp32 path: k_rb_fwd<32,D>
                        against the fp32 matrix peak)
  "roofline_onepass_bwd" : the same call one level down (C = 16), where it runs as the one-pass strip kernel k_wrb_bwds (round 4)
  "roofline_narrow_bwd" : the same call at the narrow levels (C = 4, 8: k_nrb_bwd_fused, by time the largest family of the step), one entry each
  "host"              : host_enqueue_ms -- wall time Python + autograd need to enqueue a step with no sync inside -- next to the synced time
                        of the same steps (host_over_gpu), and the cores this process may use
  "core_capped_run"   : N = 1: the timed loop again in a child process restricted to TWO host cores (`--cores 2`, affinity set before torch is
                        imported): what a rank of an 8-rank job gets on a 16-core host; vs_uncapped = its ms_per_step / this run's
  "roofline_cqt"      : tt_cqt_forward, measured in the timed steps (HBM bound, 4,688,280 algorithmic bytes per clip)
  "roofline_cqt_inv"  : tt_cqt_inverse (CQT.decode) on the same batch, measured after the timed region (training never calls it)
  "families"          : per kernel family (narrow / wide residual blocks, strided, transposed, latent GEMMs, boundary convs,
                        losses, optimizer) ms per step, achieved TFLOP/s and TB/s on ALGORITHMIC work and the fractions of the
                        fp32 matrix peak and of HBM peak -- measured with HIP events in two extra, untimed, instrumented steps
  "whole_step"        : algorithmic conv FLOPs of the step / ms_per_step against the fp32 and the bf16 matrix peaks
  "overlap"           : N > 1 only: event timestamps of one instrumented step -- how long the compute stream still had to wait for
                        the all-reduce AFTER it had finished the next batch's CQT (0 = the collective was hidden entirely)
  "skip_connections_step" : the same step with skip_connections=True (the model of BASELINE.json configs[4]), N = 1
  "fp16_train_step"   : the same step under torch.autocast(dtype=float16) -- the unmodified train.py's dtype (train.py:415) -- with the static
                        loss scale of the fp16 backward (ops.FP16_LOSS_SCALE, round 5), N = 1
  "inference_config1" : BASELINE.json configs[1] (transcribe() + reconstruct(), 32 clips) timed in the same run, N = 1: fp32 semantics
                        (no autocast; the residual levels on split fp16 operands, csrc/conv_x3.hip -- "fp32_kernels_only" is the same leg with
                        ops.X3_INFER off, "roofline_x3_fwd" / "roofline_x3n_fwd" the wide / narrow split-operand blocks against the HBM
                        peak), and under bf16 / fp16 autocast
  "cpu_baseline"      : the CPU oracle (kind "port") on a bounded sample of the SAME workload (model_complexity 2; rank 0, N = 1)
  "cpu_baseline_config0" : BASELINE.json configs[0] on the oracle: model_complexity 1, one clip, CQT forward + inverse + one step
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


def _apply_core_cap(argv):
    """`--cores K`: restrict THIS process to K cores of its current affinity set before torch (and its thread pools) is imported and
    long before anything touches the GPU -- the one-GPU proxy for what a rank of an 8-rank job on this host gets (SURVEY.md section 8e:
    ">= 6.5x hinges on ... per-step host syncs"; round-5 verdict, weak #12).  Affects the host side only."""
    for i, a in enumerate(argv):
        k = None
        if a == '--cores' and i + 1 < len(argv):
            k = int(argv[i + 1])
        elif a.startswith('--cores='):
            k = int(a.split('=', 1)[1])
        if k:
            cores = sorted(os.sched_getaffinity(0))[:k]
            os.sched_setaffinity(0, cores)
            os.environ['OMP_NUM_THREADS'] = str(len(cores))
            return len(cores)
    return None


CORE_CAP = _apply_core_cap(sys.argv[1:])

import numpy as np  # noqa: E402
import torch  # noqa: E402

N_BLOCK, M_FRAMES, SR, N_BINS = 66150, 1024, 22050, 540
SECS_PER_CLIP = 3.0
PEAK_FP32_MATRIX_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 peak
PEAK_BF16_MFMA_TFLOPS = 2500.0       # same guide: dense bf16 MFMA
PEAK_HBM_GBS = 8000.0
MEASURED_COPY_GBS = 6290.0           # same guide: what a device-to-device copy reaches on this chip
CQT_BYTES_PER_CLIP = 4688280         # 264,600 B of audio + 4,423,680 B of coefficients


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


def build_model(mc, latent, device, seed=2, skip=False):
    from timbre_trap.framework import TimbreTrap
    torch.manual_seed(seed)                       # reference experiments/train.py:88,137
    return TimbreTrap(sample_rate=SR, n_octaves=9, bins_per_octave=60, secs_per_block=3,
                      latent_size=latent, model_complexity=mc, skip_connections=skip).to(device)


def make_train_step(model, opt, world, overlap=True, autocast=True, autocast_dtype=torch.bfloat16):
    """
    Returns step(audio, target, next_audio=None) -> total loss: exactly the body of reference experiments/train.py:404-496.

    N > 1: the flat gradient goes out as ONE all-reduce issued asynchronously -- RCCL runs it on its own
    stream once the backward kernels it depends on have finished -- and, while it is in flight, the compute stream already runs
    the CQT of the NEXT batch (`model.sliCQ(next_audio)`, the `coefficients` of train.py:404 for the following step; the
    transform has no weights, so it does not depend on the update).  The compute stream then waits for the collective and
    applies clip + AdamW.  Every step still performs its two forward transforms (train.py:404 and modules.py:88).
    """
    from timbre_trap.framework import compute_consistency_loss, compute_reconstruction_loss, compute_transcription_loss
    from timbre_trap.utils import GradientSync
    from timbre_trap.utils.distributed import dist_active
    sync = GradientSync(world) if (world > 1 or dist_active()) else None        # dist_active at world 1: TTRAP_FORCE_DIST=1 (RCCL on a 1-GPU box)
    state = dict(coeffs=None, src=None)

    def step(audio, target, next_audio=None):
        next_audio = audio if next_audio is None else next_audio
        if state['coeffs'] is not None and state['src'] is audio:
            coefficients = state['coeffs']                        # transformed during the previous step's all-reduce
        else:
            coefficients = model.sliCQ(audio)
        state['coeffs'] = state['src'] = None
        # the reference runs forward, losses and backward of the step under autocast (experiments/train.py:415); with
        # ops.PRECISION == 'auto' that region is what selects the bf16 MFMA conv path (BASELINE config[2])
        with torch.autocast(device_type='cuda', dtype=autocast_dtype, enabled=autocast):
            reconstruction, latents, trn_coeffs, trn_rec, trn_scr, _ = model(audio, True)
            transcription = model.to_activations(trn_coeffs)
            n = target.size(0)

            def head(t):
                # train.py:429-441 slices the annotated part of the batch, `t[:mpe_batch_size]`.  Here every clip is annotated, and a
                # slice that keeps everything still costs autograd a zero-filled full-size gradient plus a copy per tensor on the way
                # back (0.55 ms per step for the four tensors, `tools/prof_glue.py`): same values and gradients without it.
                return t if t.size(0) == n else t[:n]
            l_rec = compute_reconstruction_loss(reconstruction, coefficients)
            l_trn = compute_transcription_loss(head(transcription), target, True)
            l_sp, l_sc = compute_consistency_loss(head(trn_rec), head(trn_scr), head(trn_coeffs))
            total = l_rec + l_trn + (l_sp + l_sc)
            opt.zero_grad()
            total.backward()
        if sync is not None:
            if overlap:
                probe = state.get('probe')
                if probe is not None:                             # instrumented step: timestamps on the compute stream
                    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                    ev[0].record()
                sync.start(opt)
                state['coeffs'], state['src'] = model.sliCQ(next_audio), next_audio
                if probe is not None:
                    ev[1].record()
                sync.finish()
                if probe is not None:
                    ev[2].record()
                    probe.append(ev)
            else:
                sync.start(opt)
                sync.finish()
        opt.step()
        return total
    step.state = state
    return step


def family_table(events, n_steps, batch, mc, latent, bf16=False):
    """
    Per kernel family: ms per step (HIP events around every forward / backward of the autograd Functions in
    timbre_trap/framework/ops.py, recorded on the launch stream during `n_steps` untimed instrumented steps), the ALGORITHMIC
    work of those calls and the achieved rates.  Algorithmic bytes count each tensor a layer must read or write once:
    residual block forward 2 tensors (x, y), backward 3 (dy, x, dx; the saved hidden activation is an implementation choice);
    (4,1) strided / transposed layers forward in + out, backward THREE tensors when the gradient arrives already gated (x and dy read,
    dx written: ops.PREGATE, round 5 -- the ELU gate no longer needs y) and four otherwise; the losses as the passes they make:
    with ops.LOSS_FUSED the forward reads its operands and writes the gradients, the backward moves nothing (round-5 verdict, weak #8:
    both were still counted by their round-4 definitions, which put one family above what the chip can copy at).  FLOPs are 2 MAC,
    backward = 2 x forward.  A family whose fraction of HBM peak exceeds the measured copy ceiling (6.29 of 8.0
    TB/s) is flagged ``exceeds_copy_ceiling`` -- a byte definition gone stale, not a fast kernel.
    """
    from timbre_trap.framework import ops
    ch = [round(c * 2 ** (mc - 1)) for c in (2, 4, 8, 16, 32)]
    hs = [540, 269, 133, 65, 31]
    T = M_FRAMES
    lat = latent if latent else 32 * 2 ** (mc - 1)
    level = {c: i for i, c in enumerate(ch)}
    fam = {}

    def add(name, ms, calls, flops, nbytes, bound):
        f = fam.setdefault(name, dict(ms_per_step=0.0, calls_per_step=0.0, gflop_per_step=0.0, gbyte_per_step=0.0, bound=bound))
        f['ms_per_step'] += ms
        f['calls_per_step'] += calls
        f['gflop_per_step'] += flops / 1e9
        f['gbyte_per_step'] += nbytes / 1e9

    for key, ev in events.items():
        times = [e_[0].elapsed_time(e_[1]) for e_ in ev]
        # a call over 2 B clips (TimbreTrap.decode_pair runs the decoder that way) counts as two calls of the per-call work below
        ms, calls = sum(times) / n_steps, sum((e_[2] / float(batch)) if (len(e_) > 2 and e_[2]) else 1.0 for e_ in ev) / n_steps
        parts = key.split('_')
        kind, direction, tag = parts[0], parts[1] if len(parts) > 1 else '', parts[-1]
        bwd = direction == 'bwd'
        if kind == 'rb':
            C = int(tag[1:]); px = hs[level[C]] * T * batch
            fl = 2.0 * 10 * C * C * px * (2 if bwd else 1)
            by = 4.0 * C * px * (3 if bwd else 2)
            add(('narrow' if C <= 8 else 'wide') + ' residual blocks ' + ('backward' if bwd else 'forward'), ms, calls, fl * calls, by * calls,
                'hbm' if C <= 8 else 'mfma')
        elif kind == 'widelevel':
            # three residual blocks per call, bf16 channel-innermost tensors: 2 (fwd) / 3 (bwd) tensors of 2 bytes per block
            C = int(tag[1:]); px = hs[level[C]] * T * batch
            fl = 3 * 2.0 * 10 * C * C * px * (2 if bwd else 1)
            by = 3 * 2.0 * C * px * (3 if bwd else 2)
            add('residual levels (3 blocks per call, bf16 channels-last) ' + ('backward' if bwd else 'forward'), ms, calls, fl * calls, by * calls, 'hbm')
        elif kind in ('sconv16', 'tconv16'):
            C = int(tag[1:]); l = level[C]
            big, small = C * hs[l] * T * batch, 2 * C * hs[l + 1] * T * batch          # elements at the C side / the 2C side
            fl = 2.0 * 4 * C * 2 * C * hs[l + 1] * T * batch * (2 if bwd else 1)
            x_el, y_el = (big, small) if kind == 'sconv16' else (small, big)            # the layer's input / output
            if not bwd:
                by = 2.0 * (x_el + y_el)
            else:                                                 # x, dy in, dx out (+ the saved output y for the gate when it is not pregated)
                by = 2.0 * (2 * x_el + y_el + (0 if ops.PREGATE else y_el))
            add('strided + transposed (4,1) layers, bf16 channels-last ' + ('backward' if bwd else 'forward'), ms, calls, fl * calls,
                by * calls, 'hbm')
        elif kind == 'skipjoin16':
            # out = y + w * e for both halves of a pair decode (5 tensor-halves forward: 2 y, e, 2 out); backward 2 g + e in, de out
            C = int(tag[1:])
            n_el = (C * hs[level[C]] if C in level else 64 * 31) * T * batch
            add('skip joins (weight x embedding + join, one pass each way)', ms, calls, 0.0, 2.0 * n_el * (2.0 if bwd else 2.5) * calls, 'hbm')
        elif kind in ('tocl16', 'toplanar'):
            C = int(tag[1:])
            n_el = C * hs[level[C]] * T * batch if C in level else 64 * 31 * T * batch
            add('fp32 planar <-> bf16 channels-last at the fp32-only layers', ms, calls, 0.0, 6.0 * n_el * calls, 'hbm')
        elif kind in ('sconv', 'tconv'):
            C = int(tag[1:]); l = level[C]
            big, small = C * hs[l] * T * batch, 2 * C * hs[l + 1] * T * batch          # elements at the C side / the 2C side
            fl = 2.0 * 4 * C * 2 * C * hs[l + 1] * T * batch * (2 if bwd else 1)
            by = 4.0 * (big + small) * (2 if bwd else 1)
            add('strided + transposed (4,1) layers ' + ('backward' if bwd else 'forward'), ms, calls, fl * calls, by * calls, 'hbm')
        elif kind == 'edge16':
            px = hs[0] * T * batch
            fl = 2.0 * 9 * 2 * ch[0] * px * (2 if bwd else 1)
            by = (4.0 * 2 + 2.0 * ch[0]) * px * (1.5 if bwd else 1)
            add('boundary 3x3 convs (2<->%d), fp32 planar <-> bf16 channels-last' % ch[0], ms, calls, fl * calls, by * calls, 'hbm')
        elif kind in ('latenc16', 'latdec16'):
            K = ch[4] * hs[4]
            fl = 2.0 * (lat + (kind == 'latdec16')) * K * T * batch * (2 if bwd else 1)
            by = (2.0 * K + 4.0 * lat) * T * batch * (2 if bwd else 1)
            add('latent heads (31,1), bf16 channels-last embeddings', ms, calls, fl * calls, by * calls, 'hbm')
        elif kind in ('latenc', 'latdec'):
            K = ch[4] * hs[4]
            fl = 2.0 * (lat + (kind == 'latdec')) * K * T * batch * (2 if bwd else 1)
            by = 4.0 * (K + lat) * T * batch * (2 if bwd else 1)
            add('latent heads (31,1) as GEMMs', ms, calls, fl * calls, by * calls, 'mfma')
        elif kind == 'conv':
            ci, co = (int(v) for v in tag.split('to'))
            px = hs[0] * T * batch
            fl = 2.0 * 9 * ci * co * px * (2 if bwd else 1)
            by = 4.0 * (ci + co) * px * (1.5 if bwd else 1)
            add('boundary 3x3 convs (2<->%d)' % ch[0], ms, calls, fl * calls, by * calls, 'hbm')
        elif kind in ('sqdiff', 'sqdiff2', 'act', 'trn'):
            px = hs[0] * T * batch                                # fp32 planes of F x T per clip: logits 2 planes, activations / targets 1
            if ops.LOSS_FUSED:
                # forward: operands in, gradient(s) out; backward: a launch that returns when the incoming scalar is 1 (no tensor pass)
                planes = {'sqdiff': (6, 0), 'sqdiff2': (12, 0), 'act': (3, 6), 'trn': (3, 0)}[kind]
            else:
                planes = {'sqdiff': (4, 6), 'sqdiff2': (8, 12), 'act': (3, 6), 'trn': (2, 3)}[kind]
            by = 4.0 * px * planes[1 if bwd else 0]
            add('losses + to_activations', ms, calls, 0.0, by * calls, 'hbm')
        elif key == 'clip_adamw':
            add('clip + AdamW (flat buffer)', ms, calls, 0.0, 0.0, 'hbm')
        elif key == 'cqt_forward':
            add('CQT forward', ms, calls, 0.0, CQT_BYTES_PER_CLIP * batch * calls, 'hbm')
    out = {}
    for name, f in sorted(fam.items(), key=lambda kv: -kv[1]['ms_per_step']):
        sec = f['ms_per_step'] * 1e-3
        tf, tb = f['gflop_per_step'] / 1e3 / sec, f['gbyte_per_step'] / 1e3 / sec
        out[name] = dict(ms_per_step=round(f['ms_per_step'], 3), calls_per_step=f['calls_per_step'], bound=f['bound'],
                         achieved_tflops=round(tf, 2), achieved_tbs=round(tb, 3), frac_hbm_peak=round(tb * 1e3 / PEAK_HBM_GBS, 4))
        if tb * 1e3 > MEASURED_COPY_GBS:
            out[name]['exceeds_copy_ceiling'] = True
            print('bench.py: family %r reads %.2f TB/s on its algorithmic bytes -- above the %.2f TB/s this chip copies at: its byte '
                  'definition is stale' % (name, tb, MEASURED_COPY_GBS / 1e3), file=sys.stderr)
        # the matrix peak of the arithmetic the step actually runs in
        if bf16:
            out[name]['frac_bf16_mfma_peak'] = round(tf / PEAK_BF16_MFMA_TFLOPS, 4)
        else:
            out[name]['frac_fp32_matrix_peak'] = round(tf / PEAK_FP32_MATRIX_TFLOPS, 4)
    out['sum_ms_per_step'] = round(sum(f['ms_per_step'] for f in fam.values()), 3)
    return out


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


def _cpu_model_name():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return ''


def cpu_baseline(mc, latent, seconds_budget=15.0, config0=False):
    """
    The oracle (CPU restatement) on a bounded sample.  config0 = False: one clip of the bench workload (same model, the step
    of train.py:404-496 = 2 forward CQTs + autoencoder step).  config0 = True: BASELINE.json configs[0] -- model_complexity 1,
    one clip, CQT forward + inverse, then one train step (which transforms the audio again, twice, like train.py).
    The conv / loss / optimizer legs are torch fp32 on `cores` threads; the CQT legs are oracle/nsgt.py: float64 NumPy
    (one 66150-point FFT + a Python loop of 540 windowed 1024-point FFTs per clip), single-threaded.
    """
    from oracle import nsgt
    from oracle.train_step import OracleTrainer, cqt_forward_torch
    threads = min(available_cores(), 64)          # beyond ~64 threads the small conv layers only lose to sync overhead
    torch.set_num_threads(threads)
    model = build_model(mc, latent, 'cpu')
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    trainer = OracleTrainer(sd, lr=1e-3)
    tab = nsgt.nsgt_tables(9, 60, SR, N_BLOCK)
    audio, target = synthetic_batch(1, 0)
    legs = dict(cqt_forward=0.0, cqt_inverse=0.0, step=0.0)

    def one():
        t0 = time.perf_counter()
        coeffs = cqt_forward_torch(audio, tab)           # train.py:404
        _ = cqt_forward_torch(audio, tab)                # the transform again inside model.encode (modules.py:88)
        t1 = time.perf_counter()
        if config0:
            nsgt.wrapper_decode(coeffs.numpy(), tab)     # configs[0]: "CQT fwd+inv"
        t2 = time.perf_counter()
        trainer.step(coeffs, target)
        t3 = time.perf_counter()
        legs['cqt_forward'] += t1 - t0
        legs['cqt_inverse'] += t2 - t1
        legs['step'] += t3 - t2
    t0 = time.perf_counter()
    one()                                                # warm-up (also the sample if the host is very slow)
    warm = time.perf_counter() - t0
    warm_legs = dict(legs)
    for k in legs:
        legs[k] = 0.0
    n, el = 0, 0.0
    t0 = time.perf_counter()
    while warm < seconds_budget and el + warm < seconds_budget and n < 5:
        one()
        n += 1
        el = time.perf_counter() - t0
    if n == 0:
        legs, el, n = warm_legs, warm, 1
    step_s = el / n
    what = ('BASELINE configs[0]: CQT forward x2 + inverse + train step' if config0 else 'train step incl. its 2 forward CQTs')
    return dict(value=SECS_PER_CLIP / step_s, unit='audio-seconds/s', cores=threads, kind='port',
                sample='%d x (%s) of 1 clip x 3 s, model_complexity=%d latent=%s, after 1 warm-up; conv/loss/AdamW legs torch fp32 on %d '
                       'threads, CQT legs float64 NumPy (Python loop over 540 bins, 1 thread); %s'
                       % (n, what, mc, latent, threads, _cpu_model_name()),
                s_per_step=step_s, legs_s={k: v / n for k, v in legs.items()})


def bench_inference(model, args, rank, world, dev, steps=None, warmup=None, emit=True, autocast=False, dtype=None):
    """BASELINE.json configs[1]: model.transcribe(audio) + model.reconstruct(audio), batch x 3 s clips (secondary line).
    autocast=True: the same calls inside torch.autocast (16-bit channels-last path of element type ``dtype``, default bfloat16) -- what a
    user who evaluates under autocast gets; the reference's evaluate.py does not, so the fp32 figure is the one that carries the 1e-4
    output bar."""
    dtype = dtype or torch.bfloat16
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    was_training = model.training
    model.eval()
    batch = 32 if args.batch == 64 else args.batch
    audio, _ = synthetic_batch(batch, rank, dev)

    def step():
        with torch.no_grad(), torch.autocast(device_type='cuda', dtype=dtype, enabled=autocast):
            act = model.transcribe(audio)
            rec = model.reconstruct(audio)
        return act, rec
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        act, rec = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    model.train(was_training)
    ms = 1000.0 * elapsed / steps
    line = dict(metric='audio-seconds/s inference throughput, transcribe()+reconstruct() (9oct x 60bpo, 3s@22.05kHz)',
                value=world * batch * SECS_PER_CLIP / (elapsed / steps), unit='audio-seconds/s', n_gpus=world,
                steps=steps, warmup=warmup, ms_per_step=ms, higher_is_better=True, scaling='weak',
                vs_baseline=None, dtype=('f16' if dtype == torch.float16 else 'bf16') if autocast else args.infer_dtype, data='synthetic',
                config=dict(workload='transcribe() + reconstruct() (each: 3 half-overlapping chunks per clip through CQT + '
                                     'encoder + decoder, Hann cross-fade; reconstruct adds the inverse CQT), model_complexity=%d '
                                     'latent=%d, %d clips x 3 s' % (args.mc, args.latent, batch),
                            global_batch=world * batch, parallelism='dp%d' % world),
                out_shapes=[list(act.shape), list(rec.shape)])
    if emit and rank == 0:
        print(json.dumps(line))
    return line


def self_launch(args, argv):
    """
    `python bench.py --gpus N` with N > 1 and no WORLD_SIZE: start the N ranks as fresh child processes (SURVEY.md section 8e:
    one process per GPU, replacing the in-process nn.DataParallel of reference experiments/train.py:166-168).  Nothing in THIS
    process may initialise the GPU: hipcc children and torch.distributed.run are started from a clean parent, which only counts
    devices (torch.cuda.device_count() does not create a HIP context on this image).  Returns the exit code.
    """
    import socket
    import subprocess
    from timbre_trap import _hip
    _hip.build()
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and os.environ.get('TTRAP_DIST_BACKEND') != 'gloo':
        print('bench.py: --gpus %d but only %d GPU(s) visible -- refusing to measure fewer GPUs than asked '
              '(TTRAP_DIST_BACKEND=gloo shares the visible device(s) between ranks for a functional test)' % (args.gpus, n_dev),
              file=sys.stderr)
        return 2
    with socket.socket() as s_:
        s_.bind(('127.0.0.1', 0))
        port = s_.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC only on this host driver (RCCL needs it)
    env.setdefault('OMP_NUM_THREADS', str(max(1, available_cores() // args.gpus)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for ln in proc.stdout.splitlines():
        if ln.startswith('{'):
            print(ln)                                      # rank 0's JSON line(s); launcher chatter goes to stderr already
        else:
            print(ln, file=sys.stderr)
    sys.stdout.flush()
    return proc.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # five warm-up steps: the caching allocator and the clocks settle over the first few 64-clip steps (measured: 2 warm-up + 8
    # timed steps read 204 ms/step where 5 + 20 read 191 ms/step and the kernel trace 192 ms/step on the same box)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=64, help='clips per GPU')
    ap.add_argument('--mc', type=int, default=2)
    ap.add_argument('--latent', type=int, default=128)
    ap.add_argument('--precision', choices=('auto', 'fp32', 'bf16x3', 'bf16', 'fp16'), default=os.environ.get('TTRAP_PRECISION', 'auto'),
                    help="auto (default) = like the reference: the train step runs under torch.autocast and takes the bf16 MFMA conv path "
                         "(BASELINE config[2]), inference runs in exact fp32; fp32 / bf16x3 / bf16 force one arithmetic everywhere")
    ap.add_argument('--mode', choices=('train', 'infer'), default='train',
                    help="train = the headline metric; infer = BASELINE config[1]: transcribe() + reconstruct() on 32 clips x 3 s")
    ap.add_argument('--skip-connections', action='store_true',
                    help='time the skip_connections=True model (BASELINE configs[4]) as the main line (profiling; the default line reports it as '
                         '"skip_connections_step")')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--timed-only', action='store_true',
                    help='skip the untimed legs (instrumented family steps, inverse CQT, inference config, CPU baselines): for rocprofv3 runs whose '
                         'kernel totals should divide by warmup + steps')
    ap.add_argument('--cores', type=int, default=None,
                    help='restrict this process to K host cores (applied at import, before torch; see _apply_core_cap)')
    ap.add_argument('--capped-run', choices=('auto', 'on', 'off'), default='auto',
                    help='N = 1: repeat the timed loop in a child process on two host cores ("core_capped_run"); auto = with the CPU baselines')
    ap.add_argument('--no-overlap', action='store_true',
                    help='N > 1: blocking all-reduce on the compute stream instead of the side-stream all-reduce overlapped with the next CQT')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))

    from timbre_trap import _hip
    from timbre_trap.framework import ops
    from timbre_trap.utils import FusedAdamW, allreduce_gradients, init_process_group_from_env
    ops.PRECISION = args.precision
    train_dtype = {'auto': 'bf16', 'fp32': 'f32', 'bf16x3': 'bf16x3', 'bf16': 'bf16', 'fp16': 'f16'}[args.precision]
    infer_dtype = {'auto': 'f32', 'fp32': 'f32', 'bf16x3': 'bf16x3', 'bf16': 'bf16', 'fp16': 'f16'}[args.precision]
    args.infer_dtype = infer_dtype
    from timbre_trap.utils.distributed import broadcast_parameters, dist_active
    import torch.distributed as dist

    # Build decision BEFORE anything initialises the GPU or the process group (init_process_group_from_env selects the device
    # for RCCL; hipcc children must not be started from a process that holds it) and identical on every rank: local rank 0 --
    # read from the environment -- calls build(), a no-op when the library is up to date and an atomic rename into place
    # otherwise; EVERY rank then passes the same barrier before dlopen.
    if int(os.environ.get('LOCAL_RANK', os.environ.get('RANK', '0'))) == 0:
        _hip.build()
    rank, world, local_rank = init_process_group_from_env()
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    # the data-parallel exchange is live with more than one rank -- or with ONE rank under TTRAP_FORCE_DIST=1 (how RCCL is exercised on
    # a 1-GPU box: same process group, broadcast, asynchronous all-reduce and stream wait; the figures of `overlap` / `allreduce_ms`
    # are then those of a one-rank communicator)
    multi = world > 1 or dist_active()
    if multi:
        dist.barrier()
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (there is no CPU fallback for the HIP path)')
    if local_rank >= torch.cuda.device_count():
        if os.environ.get('TTRAP_DIST_BACKEND') != 'gloo':
            raise SystemExit('LOCAL_RANK %d but only %d GPU(s) visible' % (local_rank, torch.cuda.device_count()))
        local_rank %= torch.cuda.device_count()       # functional test of the N > 1 path: gloo ranks sharing the box's GPU(s)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    _hip.lib()

    model = build_model(args.mc, args.latent, dev, skip=args.skip_connections)
    if args.mode == 'infer':
        return bench_inference(model, args, rank, world, dev)
    opt = FusedAdamW(model.parameters(), lr=1e-3, max_norm=10.0)
    if multi:
        broadcast_parameters(opt.flat_param)
    audio, target = synthetic_batch(args.batch, rank, dev)
    step_fn = make_train_step(model, opt, world, overlap=not args.no_overlap, autocast=args.precision == 'auto')

    def sync():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step_fn(audio, target)
    sync()
    torch.cuda.reset_peak_memory_stats()
    C = 16 * 2 ** (args.mc - 1)
    key, key16, keyb = 'resblock_fwd_C%d' % C, 'wide_rb_fwd_C%d' % C, 'wide_rb_bwd_C%d' % C
    keyb_half = 'wide_rb_bwd_C%d' % (C // 2)              # the level below: its backward runs as the one-pass strip kernel
    keyb_narrow = ['wide_rb_bwd_C%d' % c for c in (C // 8, C // 4)]      # the narrow levels: k_nrb_bwd_fused (the largest family of the step)
    _hip.EVENT_KEYS = {key, key16, keyb, keyb_half, *keyb_narrow, 'cqt_forward'}   # the timed region brackets only the roofline calls
    _hip.EVENT_LOG = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        total = step_fn(audio, target)
    sync()
    elapsed = time.perf_counter() - t0
    peak_gb = torch.cuda.max_memory_allocated() / 1e9            # weights + optimizer + every tensor saved for backward
    events = _hip.EVENT_LOG
    _hip.EVENT_LOG = None
    rank_ms = 1000.0 * elapsed / args.steps
    per_rank_ms = [rank_ms]
    if multi:
        # RCCL moves device tensors, gloo (functional tests of the N > 1 path on one GPU) host tensors
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == 'nccl' else 'cpu')
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        per_rank_ms = [1000.0 * float(g.item()) / args.steps for g in gathered]
        elapsed = max(float(g.item()) for g in gathered)
    ms = 1000.0 * elapsed / args.steps
    value = world * args.batch * SECS_PER_CLIP / (elapsed / args.steps)

    # ---- the host side of a step (round-5 verdict, weak #12): how long Python + autograd need to ENQUEUE a step, with no sync inside --
    # wall time until the last launch of `n_enq` steps has returned, next to the synced wall time of the same steps.  host_over_gpu << 1:
    # the host stays ahead of the device (the launch queue absorbs the difference); -> 1: the step is host-bound.  Every rank runs it.
    n_enq = 4
    sync()
    t_e0 = time.perf_counter()
    for _ in range(n_enq):
        step_fn(audio, target)
    t_e1 = time.perf_counter()
    torch.cuda.synchronize()
    t_e2 = time.perf_counter()
    host_enqueue_ms = 1000.0 * (t_e1 - t_e0) / n_enq
    host_info = dict(host_enqueue_ms=host_enqueue_ms, synced_ms=1000.0 * (t_e2 - t_e0) / n_enq, host_over_gpu=(t_e1 - t_e0) / (t_e2 - t_e0),
                     steps=n_enq, host_cores=available_cores(),
                     note='wall time until the last launch of the steps returned (no sync inside) / the same steps synced; a ratio near 1 '
                          'would mean the host, not the GPU, paces the step')

    # ---- untimed, instrumented legs (every rank runs them so collectives stay matched; rank 0 reports) ----
    fam_events, n_inst = {}, (0 if args.timed_only else 2)
    _hip.EVENT_KEYS = None
    _hip.EVENT_LOG = fam_events
    for _ in range(n_inst):
        step_fn(audio, target)
    torch.cuda.synchronize()
    _hip.EVENT_LOG = None
    allreduce_ms = overlap_info = None
    if multi and not args.no_overlap:
        # did the next batch's CQT really run beside the all-reduce?  Compute-stream timestamps of two instrumented steps:
        # [start of the exchange] -> [CQT enqueued and finished] -> [the stream's wait for the collective released]
        step_fn.state['probe'] = []
        for _ in range(2):
            step_fn(audio, target)
        torch.cuda.synchronize()
        pr = step_fn.state.pop('probe')
        cq = sum(e[0].elapsed_time(e[1]) for e in pr) / len(pr)
        wt = sum(e[1].elapsed_time(e[2]) for e in pr) / len(pr)
        overlap_info = dict(cqt_on_compute_stream_ms=cq, wait_for_collective_after_cqt_ms=wt, backend=dist.get_backend(), world_size=world,
                            communicator='one-rank communicator (TTRAP_FORCE_DIST=1): the collective is a no-op -- these figures show co-existence and stream '
                                         'ordering, not bandwidth' if world == 1 else '%d ranks' % world,
                            note='RCCL runs the collective on its own stream; wait << allreduce_ms means it was hidden behind the CQT '
                                 '(gloo blocks the host instead: the figures are then host-side)')
    if multi:
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a0.record()
        for _ in range(10):
            allreduce_gradients(opt.flat_grad, world)
        a1.record()
        torch.cuda.synchronize()
        allreduce_ms = a0.elapsed_time(a1) / 10
    inv_events = {}
    with torch.no_grad():
      if not args.timed_only:
        coeffs = model.sliCQ(audio)
        for _ in range(3):
            model.sliCQ.decode(coeffs)
        _hip.EVENT_KEYS = {'cqt_inverse'}
        _hip.EVENT_LOG = inv_events
        for _ in range(20):
            model.sliCQ.decode(coeffs)
        torch.cuda.synchronize()
        _hip.EVENT_LOG = None
        _hip.EVENT_KEYS = None

    if rank == 0:
        def avg_ms(ev):
            # average duration of a call over args.batch clips: a call over 2 B clips (TimbreTrap.decode_pair: the decoder) counts as two
            times = [e_[0].elapsed_time(e_[1]) for e_ in ev]
            units = sum((e_[2] / float(args.batch)) if (len(e_) > 2 and e_[2]) else 1.0 for e_ in ev)
            return sum(times) / units, int(round(units))
        # the fused residual block at the widest level (C = 16*mc channels, H = 65 rows)
        roof = None
        if events.get(key):
            a_ms, n_l = avg_ms(events[key])
            flops = 2.0 * (9 * C * C + C * C) * args.batch * 65 * M_FRAMES
            ach = flops / (a_ms * 1e-3) / 1e12
            traffic, traffic_source = None, None
            for name in ('r02_d_pmc_rb_fwd_C32.json', 'r02_pmc_rb_fwd_C32.json', 'r01_f_pmc_rb_fwd_C32.json'):
                pmc = os.path.join(ROOT, 'profiles', name)
                if C == 32 and args.batch == 64 and args.precision == 'fp32' and os.path.exists(pmc):
                    traffic = json.load(open(pmc))['traffic_bytes_corrected']
                    traffic_source = 'profiles/%s (rocprofv3 --pmc passes of this kernel at this shape: FETCH_SIZE x2 + WRITE_SIZE; not re-measured in this run)' % name
                    break
            roof = dict(kernel='k_rb_fwd<%d,D> (fused ResidualConv2dBlock forward, C=%d, H=65; same MFMA main loop as the '
                               'data-gradient kernel k_conv_mfma)' % (C, C),
                        bound='mfma', achieved=ach, peak=PEAK_FP32_MATRIX_TFLOPS, unit='TFLOP/s',
                        frac=ach / PEAK_FP32_MATRIX_TFLOPS, traffic=traffic, traffic_source=traffic_source, algorithmic_flops=flops,
                        algorithmic_bytes=2.0 * 4 * C * args.batch * 65 * M_FRAMES, launches=n_l, avg_ms=a_ms)
        if events.get(key16):
            # bf16-storage path: the same block reads x and writes y as bf16 -- 16x the matrix rate, so HBM is the bound
            a_ms, n_l = avg_ms(events[key16])
            nbytes = 2.0 * 2 * C * args.batch * 65 * M_FRAMES
            flops = 2.0 * (9 * C * C + C * C) * args.batch * 65 * M_FRAMES
            gbs = nbytes / (a_ms * 1e-3) / 1e9
            traffic, traffic_source = None, None
            pmc = os.path.join(ROOT, 'profiles', 'r04_pmc_wrb_fwd_C32.json')
            if C == 32 and args.batch == 64 and os.path.exists(pmc):
                traffic = json.load(open(pmc))['traffic_bytes_corrected']
                traffic_source = 'profiles/r04_pmc_wrb_fwd_C32.json (rocprofv3 --pmc passes of this kernel at this shape, round 4: FETCH_SIZE x2 + WRITE_SIZE, mean of the three dilations; not re-measured in this run)'
            roof = dict(kernel='k_wrb_conv<%d,D,0> (fused ResidualConv2dBlock forward, bf16 channel-innermost storage, C=%d, H=65)' % (C, C),
                        bound='hbm', achieved=gbs, peak=PEAK_HBM_GBS, unit='GB/s', frac=gbs / PEAK_HBM_GBS, traffic=traffic,
                        traffic_source=traffic_source, algorithmic_bytes=nbytes, algorithmic_flops=flops,
                        achieved_tflops=flops / (a_ms * 1e-3) / 1e12, frac_bf16_mfma_peak=flops / (a_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                        launches=n_l, avg_ms=a_ms,
                        note='algorithmic bytes = x read + y written once (bf16); the hidden activation saved for backward (+50 %) is an '
                             'implementation choice and is not counted')
        roof_fwd = roof
        if events.get(keyb):
            # the by-time dominant call of the bf16 step: the residual-block backward at the widest level (18 calls per step)
            a_ms, n_l = avg_ms(events[keyb])
            if ops.LEVEL_BWD:                                     # one event = tt_wide_level_bwd = the three blocks of a level (one reduce launch)
                a_ms, n_l = a_ms / 3.0, n_l * 3
            nbytes = 3.0 * 2 * C * args.batch * 65 * M_FRAMES
            flops = 2.0 * 2.0 * (9 * C * C + C * C) * args.batch * 65 * M_FRAMES
            gbs = nbytes / (a_ms * 1e-3) / 1e9
            traffic, traffic_source, pmc_note = None, None, None
            pmc = os.path.join(ROOT, 'profiles', 'r05_pmc_wrb_bwd_C32.json')       # round 5: k_wrb_dxw with the halo-free x tile
            if C == 32 and args.batch == 64 and os.path.exists(pmc):
                pj = json.load(open(pmc))
                traffic = pj['traffic_bytes_corrected']
                pmc_note = pj.get('summary')
                traffic_source = 'profiles/r05_pmc_wrb_bwd_C32.json (rocprofv3 --pmc passes of the kernels of this call at this shape, round 5: FETCH_SIZE x2 + WRITE_SIZE, summed, mean of the three dilations; round 3 / 4: 2160-2190 MB; not re-measured in this run)'
            roof = dict(kernel='tt_wide_rb_bwd at C=%d, H=65 (ResidualConv2dBlock backward, bf16 channel-innermost storage): k_wrb_bwd_a<%d> + '
                               'k_wrb_dxw<%d,D,8,32> (data + weight gradient in one pass, halo-free x tile; the first block of a level leaves dx gated for the layer in front) + k_wrb_reduce<%d>; the by-time dominant call of the step '
                               '(the one-pass strip kernel, default one level down, loses here inside the step: roofline_onepass_bwd, DESIGN.md section 7)' % (C, C, C, C),
                        bound='hbm', achieved=gbs, peak=PEAK_HBM_GBS, unit='GB/s', frac=gbs / PEAK_HBM_GBS, traffic=traffic,
                        traffic_source=traffic_source, algorithmic_bytes=nbytes, algorithmic_flops=flops,
                        achieved_tflops=flops / (a_ms * 1e-3) / 1e12, frac_bf16_mfma_peak=flops / (a_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                        launches=n_l, avg_ms=a_ms, ms_per_step=a_ms * n_l / args.steps, pmc=pmc_note,
                        note='algorithmic bytes = dy and x read, dx written once (bf16); the two passes also read the saved hidden '
                             'activation, write and re-read dL/d(conv1 pre-activation) and read dy twice -- see traffic')
        roof_onepass = None
        if events.get(keyb_half) and C == 32 and train_dtype in ('bf16', 'f16'):
            a_ms, n_l = avg_ms(events[keyb_half])
            if ops.LEVEL_BWD:
                a_ms, n_l = a_ms / 3.0, n_l * 3
            Ch, Hh = C // 2, 133
            nbytes = 3.0 * 2 * Ch * args.batch * Hh * M_FRAMES
            gbs = nbytes / (a_ms * 1e-3) / 1e9
            traffic = traffic_source = None
            pmc = os.path.join(ROOT, 'profiles', 'r06_pmc_bwds_C16.json')         # round 6: re-measured on the round-5 strip kernel (x rows in their own image)
            if args.batch == 64 and os.path.exists(pmc):
                pj = json.load(open(pmc))
                traffic, traffic_source = pj['traffic_bytes_corrected'], 'profiles/r06_pmc_bwds_C16.json (rocprofv3 --pmc passes of this kernel at this shape, round 6: FETCH_SIZE x2 + WRITE_SIZE, mean of the three dilations; not re-measured in this run)'
            roof_onepass = dict(kernel='tt_wide_rb_bwd at C=%d, H=%d: k_wrb_bwds<%d,D,8,32> (one-pass strip backward: h1, dy, x in, dx out; dL/d(conv1 '
                                       'pre-activation) in an LDS ring) + k_wrb_reduce<%d>' % (Ch, Hh, Ch, Ch),
                                bound='hbm', achieved=gbs, peak=PEAK_HBM_GBS, unit='GB/s', frac=gbs / PEAK_HBM_GBS, traffic=traffic,
                                traffic_source=traffic_source, algorithmic_bytes=nbytes, launches=n_l, avg_ms=a_ms,
                                ms_per_step=a_ms * n_l / args.steps,
                                note='average per block (the event brackets tt_wide_level_bwd: three blocks + one reduce launch); algorithmic bytes = dy and x read, dx written once (the same definition as `roofline`); the kernel also reads '
                                     'the saved hidden activation: 4 tensors of traffic where the per-stage kernels move 7')
        roof_narrow = []
        if train_dtype in ('bf16', 'f16'):
            pmc_txt = os.path.join(ROOT, 'profiles', 'r05_pmc_bwd_narrow.txt')
            pmc_rows = {}
            if os.path.exists(pmc_txt):
                import re
                for ln in open(pmc_txt):
                    m_ = re.match(r'k_nrb_bwd_fused<(\d+), (\d+), \w+>\s+traffic\s+([0-9.]+) MB', ln)
                    if m_:
                        pmc_rows.setdefault(int(m_.group(1)), []).append(float(m_.group(3)) * 1e6)
            for kn in keyb_narrow:
                if not events.get(kn):
                    continue
                Cn = int(kn.rsplit('C', 1)[1])
                Hn = {C // 8: 540, C // 4: 269}[Cn]
                a_ms, n_l = avg_ms(events[kn])
                if ops.LEVEL_BWD:
                    a_ms, n_l = a_ms / 3.0, n_l * 3
                nbytes = 3.0 * 2 * Cn * args.batch * Hn * M_FRAMES
                gbs = nbytes / (a_ms * 1e-3) / 1e9
                traffic = traffic_source = None
                if args.batch == 64 and args.mc == 2 and pmc_rows.get(Cn):
                    traffic = sum(pmc_rows[Cn]) / len(pmc_rows[Cn])
                    traffic_source = ('profiles/r05_pmc_bwd_narrow.txt (rocprofv3 --pmc passes of k_nrb_bwd_fused<%d,D> at this shape: FETCH_SIZE x2 + WRITE_SIZE, '
                                      'mean of the three dilations; not re-measured in this run)' % Cn)
                roof_narrow.append(dict(kernel='tt_wide_rb_bwd at C=%d, H=%d: k_nrb_bwd_fused<%d,D> (the whole backward of a narrow block in one pass: h1, dy, x in, dx out) '
                                               '+ k_nrb_reduce<%d>' % (Cn, Hn, Cn, Cn),
                                        bound='hbm', achieved=gbs, peak=PEAK_HBM_GBS, unit='GB/s', frac=gbs / PEAK_HBM_GBS, traffic=traffic,
                                        traffic_source=traffic_source, algorithmic_bytes=nbytes, launches=n_l, avg_ms=a_ms,
                                        ms_per_step=a_ms * n_l / args.steps,
                                        note='average per block (the event brackets tt_wide_level_bwd: three blocks + one reduce launch); algorithmic bytes = dy and x read, '
                                             'dx written once (the definition of `roofline`); the kernel also reads the saved hidden activation: 4 tensors'))
        cqt = cqt_inv = None
        if events.get('cqt_forward'):
            a_ms, n_l = avg_ms(events['cqt_forward'])
            gbs = args.batch * CQT_BYTES_PER_CLIP / (a_ms * 1e-3) / 1e9
            cqt = dict(kernel='tt_cqt_forward', bound='hbm', achieved=gbs, peak=PEAK_HBM_GBS, unit='GB/s',
                       frac=gbs / PEAK_HBM_GBS, avg_ms=a_ms, launches=n_l, clips=args.batch)
        if inv_events.get('cqt_inverse'):
            a_ms, n_l = avg_ms(inv_events['cqt_inverse'])
            gbs = args.batch * CQT_BYTES_PER_CLIP / (a_ms * 1e-3) / 1e9
            cqt_inv = dict(kernel='tt_cqt_inverse (CQT.decode incl. the batch infinity norm)', bound='hbm', achieved=gbs, peak=PEAK_HBM_GBS,
                           unit='GB/s', frac=gbs / PEAK_HBM_GBS, avg_ms=a_ms, launches=n_l, clips=args.batch,
                           note='measured after the timed region; the train step never calls the inverse transform')
        families = family_table(fam_events, n_inst, args.batch, args.mc, args.latent, bf16=train_dtype == 'bf16') if n_inst else None
        conv_flops = {1: 41.4e9, 2: 168.7e9}.get(args.mc)          # SURVEY.md section 8d, fwd + bwd per clip (latent 32 / 128)
        whole = None
        if conv_flops:
            tf = conv_flops * args.batch / (ms * 1e-3) / 1e12
            whole = dict(algorithmic_conv_flops_per_step=conv_flops * args.batch, achieved_tflops=tf,
                         frac_fp32_matrix_peak=tf / PEAK_FP32_MATRIX_TFLOPS, frac_bf16_mfma_peak=tf / PEAK_BF16_MFMA_TFLOPS)
        infer = None
        if world == 1 and not args.timed_only:            # BASELINE configs[1], measured in the same run (secondary figure)
            full = bench_inference(model, args, rank, world, dev, steps=10, warmup=2, emit=False)
            infer = {k: full[k] for k in ('value', 'unit', 'ms_per_step', 'steps', 'warmup', 'dtype')}
            infer['workload'] = full['config']['workload']
            if args.precision == 'auto' and ops.X3_INFER:
                # no autocast, no grad: the wide levels (C = 16, 32) run on split fp16 operands (csrc/conv_x3.hip: fp32-class results,
                # tests/test_gpu_x3.py; model-level bar 1e-4 in tests/test_gpu_model.py), the rest on the fp32 kernels
                infer['dtype'] = 'f32 (residual levels of all four widths: f16 hi/lo pairs, three f16 MFMA products, f32 accumulate)'
                ops.X3_INFER = False
                plain = bench_inference(model, args, rank, world, dev, steps=10, warmup=2, emit=False)
                ops.X3_INFER = True
                infer['fp32_kernels_only'] = {k: plain[k] for k in ('value', 'unit', 'ms_per_step', 'dtype')}
                # its own roofline entry: the split-operand block at the widest level, HIP events around every launch of two more
                # (untimed) inference steps
                x3_events = {}
                nkeys = ['x3n_rb_fwd_C%d' % c for c in (2 * args.mc, 4 * args.mc) if c in ops.X3N_CHANNELS]
                _hip.EVENT_KEYS = {'x3_rb_fwd_C%d' % (16 * args.mc), *nkeys}
                _hip.EVENT_LOG = x3_events
                bench_inference(model, args, rank, world, dev, steps=2, warmup=0, emit=False)
                _hip.EVENT_LOG = None
                _hip.EVENT_KEYS = None
                key = 'x3_rb_fwd_C%d' % (16 * args.mc)
                if x3_events.get(key):
                    a_ms, n_l = avg_ms(x3_events[key])
                    Bx, Cx, Hx, Tx = ops.X3_SHAPES[key]
                    nbytes = 2 * Bx * Cx * Hx * Tx * 4                  # x read + y written, 4 bytes per element (two halves)
                    gbs = nbytes / (a_ms * 1e-3) / 1e9
                    traffic, traffic_source = None, None
                    pmc = os.path.join(ROOT, 'profiles', 'r04_pmc_x3_C32.json')
                    if Cx == 32 and os.path.exists(pmc):
                        pj = json.load(open(pmc))
                        traffic = pj['traffic_bytes_corrected'] * (Bx * Hx * Tx) / pj['pixels']
                        traffic_source = 'profiles/r04_pmc_x3_C32.json (rocprofv3 --pmc passes at B 64 x H 65 x T 1024, scaled by the pixel count; FETCH_SIZE x2 + WRITE_SIZE; not re-measured in this run)'
                    infer['roofline_x3_fwd'] = dict(
                        kernel='tt_x3_rb_fwd at C=%d, B=%d, H=%d, T=%d: k_x3_conv<%d,D> (fused ResidualConv2dBlock forward on fp16 hi/lo pairs, '
                               'dilation 1 / 2 / 3 averaged)' % (Cx, Bx, Hx, Tx, Cx),
                        bound='hbm', achieved=gbs, peak=PEAK_HBM_GBS, unit='GB/s', frac=gbs / PEAK_HBM_GBS, traffic=traffic,
                        traffic_source=traffic_source, algorithmic_bytes=nbytes, launches=n_l, avg_ms=a_ms,
                        note='algorithmic bytes = x read + y written at 4 bytes per element (the x3 layout is as wide as fp32)')
                # round 5: the narrow levels on the same arithmetic, lane = pixel (k_x3n_conv); fp32 planar or split tensors, 4 bytes per element either way
                narrow = []
                for nk in nkeys:
                    if x3_events.get(nk):
                        a_ms, n_l = avg_ms(x3_events[nk])
                        Bx, Cx, Hx, Tx = ops.X3_SHAPES[nk]
                        nbytes = 2 * Bx * Cx * Hx * Tx * 4
                        gbs = nbytes / (a_ms * 1e-3) / 1e9
                        narrow.append(dict(kernel='tt_x3n_rb_fwd at C=%d, B=%d, H=%d, T=%d: k_x3n_conv<%d,D,..> (dilation 1 / 2 / 3 averaged)' % (Cx, Bx, Hx, Tx, Cx),
                                           bound='hbm', achieved=gbs, peak=PEAK_HBM_GBS, unit='GB/s', frac=gbs / PEAK_HBM_GBS, traffic=None,
                                           algorithmic_bytes=nbytes, launches=n_l, avg_ms=a_ms))
                infer['roofline_x3n_fwd'] = narrow or None
            if args.precision == 'auto':
                full16 = bench_inference(model, args, rank, world, dev, steps=10, warmup=2, emit=False, autocast=True)
                infer['under_autocast'] = {k: full16[k] for k in ('value', 'unit', 'ms_per_step', 'dtype')}
                # the reference's own autocast dtype (train.py:415 takes torch's default, float16): outputs within ~1.4e-3 of fp32
                # (bf16: ~1e-2; profiles/r04_fp16_vs_bf16.txt)
                fullh = bench_inference(model, args, rank, world, dev, steps=10, warmup=2, emit=False, autocast=True, dtype=torch.float16)
                infer['under_autocast_fp16'] = {k: fullh[k] for k in ('value', 'unit', 'ms_per_step', 'dtype')}
        fp32_step = None
        if world == 1 and not args.timed_only and train_dtype != 'f32':
            # the exact-fp32 train step (every parity test's arithmetic), measured in the same run: secondary figure
            prev = ops.PRECISION
            ops.PRECISION = 'fp32'
            f_step = make_train_step(model, opt, world, overlap=False, autocast=False)
            for _ in range(2):
                f_step(audio, target)
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats()
            t1 = time.perf_counter()
            for _ in range(4):
                f_step(audio, target)
            torch.cuda.synchronize()
            f_ms = 1000.0 * (time.perf_counter() - t1) / 4
            ops.PRECISION = prev
            fp32_step = dict(ms_per_step=f_ms, value=args.batch * SECS_PER_CLIP / (f_ms * 1e-3), unit='audio-seconds/s', dtype='f32',
                             steps=4, warmup=2, peak_memory_gb=torch.cuda.max_memory_allocated() / 1e9, note='same step with ops.PRECISION = fp32 (no autocast): bit-exact fp32 MFMA path')
        fp16_step = None
        if world == 1 and not args.timed_only and args.precision == 'auto':
            # the reference's OWN autocast dtype (train.py:415: torch.autocast('cuda') = float16) with the static loss scale of round 5
            # (ops.FP16_LOSS_SCALE): same kernels compiled with fp16 elements; gradients 7x closer to fp32 than bf16's at this batch
            # (profiles/r05_fp16_vs_bf16.txt) -- secondary figure, the headline keeps the bf16 path BASELINE configs[2] names
            h_step = make_train_step(model, opt, world, overlap=False, autocast=True, autocast_dtype=torch.float16)
            for _ in range(2):
                h_step(audio, target)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(6):
                h_step(audio, target)
            torch.cuda.synchronize()
            h_ms = 1000.0 * (time.perf_counter() - t1) / 6
            fp16_step = dict(ms_per_step=h_ms, value=args.batch * SECS_PER_CLIP / (h_ms * 1e-3), unit='audio-seconds/s', dtype='f16', steps=6, warmup=2,
                             loss_scale=ops.FP16_LOSS_SCALE, skipped_steps=int(opt.skipped),
                             note='the same train step under torch.autocast(dtype=float16) -- the unmodified train.py\'s dtype -- with the static loss scale')
        skip_step = None
        if world == 1 and not args.timed_only and args.precision == 'auto':
            # BASELINE configs[4] trains with skip_connections=True: the same step with the five weighted skip joins (cl16 joins on the
            # device: tt_scaled_add16 / tt_dot16), secondary figure
            s_model = build_model(args.mc, args.latent, dev, skip=True)
            s_opt = FusedAdamW(s_model.parameters(), lr=1e-3, max_norm=10.0)
            s_step = make_train_step(s_model, s_opt, world, overlap=False, autocast=True)
            for _ in range(2):
                s_step(audio, target)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(6):
                s_step(audio, target)
            torch.cuda.synchronize()
            s_ms = 1000.0 * (time.perf_counter() - t1) / 6
            skip_step = dict(ms_per_step=s_ms, value=args.batch * SECS_PER_CLIP / (s_ms * 1e-3), unit='audio-seconds/s', dtype='bf16', steps=6,
                             warmup=2, note='the same train step with skip_connections=True (BASELINE configs[4] model)')
            del s_model, s_opt, s_step
        capped = None
        want_capped = args.capped_run == 'on' or (args.capped_run == 'auto' and not args.no_cpu_baseline)
        if world == 1 and not args.timed_only and want_capped and CORE_CAP is None and args.precision == 'auto' and available_cores() > 2:
            # the same timed loop in a CHILD process restricted to two host cores from its first instruction (`--cores 2`: affinity set
            # before torch is imported; this parent only waits) -- what a rank gets when eight of them share this box's cores
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), '--cores', '2', '--timed-only', '--steps', str(max(args.steps, 10)), '--warmup', '3',
                   '--batch', str(args.batch), '--mc', str(args.mc), '--latent', str(args.latent)]
            try:
                r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
                child = next((json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith('{')), None)
            except (subprocess.TimeoutExpired, ValueError):
                child = None
            if child:
                capped = dict(cores=child['host']['host_cores'], ms_per_step=child['ms_per_step'], vs_uncapped=child['ms_per_step'] / ms,
                              host_enqueue_ms=child['host']['host_enqueue_ms'], host_over_gpu=child['host']['host_over_gpu'],
                              steps=child['steps'], warmup=child['warmup'],
                              note='python bench.py --cores 2 --timed-only as a child process: the whole train step with the host side on two cores')
        base = base0 = None
        if not args.no_cpu_baseline and not args.timed_only and world == 1:
            base = cpu_baseline(args.mc, args.latent)
            base0 = cpu_baseline(1, None, config0=True)
        line = dict(metric='audio-seconds/s training throughput (9oct x 60bpo, 3s@22.05kHz)', value=value,
                    unit='audio-seconds/s', n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=ms,
                    higher_is_better=True, scaling='weak', vs_baseline=None, dtype=train_dtype, data='synthetic',
                    config=dict(workload='full train step (CQT x2 + AE fwd/bwd with consistency + 3 losses + clip + AdamW), '
                                         'model_complexity=%d latent=%d%s, %d clips x 3 s per GPU%s'
                                         % (args.mc, args.latent, ' skip_connections=True' if args.skip_connections else '', args.batch, ', under torch.autocast like reference experiments/train.py:415 '
                                            '(bf16 MFMA conv path of BASELINE config[2]; fp32 master weights, losses and optimizer)'
                                            if args.precision == 'auto' else ''),
                                global_batch=world * args.batch, parallelism='dp%d' % world),
                    roofline=roof, roofline_fwd=roof_fwd if roof_fwd is not roof else None, roofline_onepass_bwd=roof_onepass,
                    roofline_narrow_bwd=roof_narrow or None, host=host_info, core_capped_run=capped, roofline_cqt=cqt, roofline_cqt_inv=cqt_inv, families=families, whole_step=whole,
                    peak_memory_gb=peak_gb, inference_config1=infer, fp32_train_step=fp32_step, fp16_train_step=fp16_step, skip_connections_step=skip_step, per_rank_ms=per_rank_ms, allreduce_ms=allreduce_ms, overlap=overlap_info, cpu_baseline=base, cpu_baseline_config0=base0,
                    final_loss=float(total.detach()))
        print(json.dumps(line))
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
