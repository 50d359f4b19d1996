"""
Pin the CQT against ``cqt_pytorch`` -- the one thing this repository cannot do in its build container.

The reference obtains every CQT value from the third-party package ``cqt_pytorch`` (archinetai/cqt-pytorch, unpinned in
reference requirements.txt:15; constructed at reference timbre_trap/framework/cqtwrapper.py:31-35).  It is not installed in
this image and there is no network, so the transform here is PARITY UNPINNED (oracle/nsgt.py).  Run this script on any
machine where ``import cqt_pytorch`` works:

    python tools/pin_cqt.py                 # no-op (exit 0) when cqt_pytorch is not importable

It then
  1. records ``cqt_pytorch.CQT(num_octaves=9, num_bins_per_octave=60, sample_rate=22050, block_length=66150,
     power_of_2_length=True)`` on closed-form audio (a two-block chirp + clicks + seeded noise): encode() coefficients at a
     strided subset of (bin, frame) positions, the decode(encode()) round trip, the package version and its registered
     buffers' shapes -> tests/golden/cqt_pytorch_pin.npz (inputs/outputs only; no source);
  2. searches the convention space of timbre_trap.framework.nsgt_plan.NSGTConventions (window family x length rounding x
     centre rounding x crop alignment -- incl. the HYPOTHESIS 'centre_minus_M' -- x bandwidth from bin k / k + 1 (HYPOTHESIS),
     then dual rule x frame-operator diagonal 'positive' / 'mirrored' (HYPOTHESIS)) with the float64 oracle and prints the
     combination(s) that reproduce the recorded coefficients to 1e-6, or the closest one with its error -- the default to set
     in nsgt_plan.DEFAULT_CONVENTIONS.  Conventions whose windows leave the positive half-spectrum are evaluated on the full
     spectrum with wrapped indices (oracle/nsgt_dense.WrappedNSGT); the device tables cannot express those yet, and the report
     says so when one of them is the match.
tests/test_cqt_pin.py compares the oracle (CPU) and the HIP transform (GPU) with the fixture whenever the file exists.
"""

import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'timbre-trap_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)
OUT = os.path.join(ROOT, 'tests', 'golden', 'cqt_pytorch_pin.npz')
N, SR = 66150, 22050
BIN_STRIDE, FRAME_STRIDE = 7, 13


def pin_audio():
    """(1, 1, 2N) float32: exponential chirp 30 Hz -> 10 kHz, three clicks, low-level seeded noise (closed form + one seed)."""
    t = np.arange(2 * N, dtype=np.float64) / SR
    dur = t[-1]
    f0, f1 = 30.0, 10000.0
    phase = 2 * np.pi * f0 * dur / np.log(f1 / f0) * (np.exp(t / dur * np.log(f1 / f0)) - 1.0)
    x = 0.6 * np.sin(phase)
    for pos in (1000, N - 3, N + 40000):
        x[pos] += 0.9
    x += 0.01 * np.random.RandomState(7).standard_normal(2 * N)
    x /= np.abs(x).max()
    return x.astype(np.float32)[None, None]


def main():
    try:
        import cqt_pytorch
        import torch
    except ImportError as e:
        print('cqt_pytorch is not importable here (%s): nothing recorded, nothing to compare -- the CQT stays parity-unpinned.' % e)
        return 0
    x = pin_audio()
    ref = cqt_pytorch.CQT(num_octaves=9, num_bins_per_octave=60, sample_rate=SR, block_length=N, power_of_2_length=True)
    with torch.no_grad():
        c = ref.encode(torch.from_numpy(x))                                   # (1, 1, 540, 2048) complex
        back = ref.decode(c)
    c = c.numpy()
    buffers = {'buffer.' + k: np.array(v.shape, dtype=np.int64) for k, v in ref.named_buffers()}
    np.savez_compressed(OUT, audio_seed=np.array([7]), coeff_sub=c[0, 0, ::BIN_STRIDE, ::FRAME_STRIDE].astype(np.complex64),
                        coeff_absmax=np.array(np.abs(c).max()), coeff_shape=np.array(c.shape), roundtrip=back.numpy().astype(np.float32)[0, 0, ::97],
                        max_window_length=np.array(getattr(ref, 'max_window_length', -1)), block_length=np.array(getattr(ref, 'block_length', -1)),
                        version=np.array(str(getattr(cqt_pytorch, '__version__', 'unknown'))), **buffers)
    print('recorded', OUT, os.path.getsize(OUT), 'bytes')
    return search(x, c, back.numpy().astype(np.float64)[0, 0])


def analysis_space():
    for window, lr, cr, crop, bw in itertools.product(('hann_periodic', 'hann_symmetric'), ('round', 'floor', 'ceil'), ('round', 'floor', 'ceil'),
                                                      ('centred', 'window_start', 'centre_minus_M'), (0, 1)):
        yield dict(window=window, length_rounding=lr, centre_rounding=cr, crop_alignment=crop, bandwidth_bin=bw)


def transform_for(kw):
    """(transform, note): the dense oracle where the windows stay inside the positive half-spectrum (what the device tables can hold),
    the wrapped full-spectrum one otherwise."""
    from oracle.nsgt_dense import DenseNSGT, WrappedNSGT
    try:
        return DenseNSGT(9, 60, SR, N, conventions=kw), ''
    except ValueError:
        return WrappedNSGT(9, 60, SR, N, conventions=kw), 'windows wrap: NOT expressible by the device tables yet'


def search(x, c, wantb, stride=(BIN_STRIDE, FRAME_STRIDE), space=None):
    """Rank the convention space (``space``: an iterable of convention dicts, default the whole analysis space) against recorded
    coefficients ``c`` (1, 1, F, T) and the recorded round trip ``wantb`` (samples)."""
    want = c[0, 0, ::stride[0], ::stride[1]]
    scale = np.abs(c).max()
    results = []
    for kw in (analysis_space() if space is None else space):
        try:
            t, note = transform_for(kw)
            got = t.encode(x.astype(np.float64))[0, 0, ::stride[0], ::stride[1]]
        except ValueError as e:
            results.append((np.inf, kw, str(e)))
            continue
        results.append((float(np.abs(got - want).max() / scale), kw, note))
    results.sort(key=lambda r: r[0])
    print('analysis conventions ranked by max |oracle - cqt_pytorch| / max |cqt_pytorch|:')
    for err, kw, note in results[:6]:
        print('  %.3e  %s %s' % (err, kw, note))
    if results[0][0] < 1e-6:
        print('MATCH: set nsgt_plan.DEFAULT_CONVENTIONS to', results[0][1], '(then compare decode() for the dual rule: tests/test_cqt_pin.py)')
    else:
        print('no combination reproduces cqt_pytorch to 1e-6: the convention space needs another switch (closest above).')
    # synthesis side: with the best analysis conventions, which dual-window rule reproduces decode(encode(x))?
    best = results[0][1]
    duals = []
    rules = [dict(dual='additive', dual_eps=1e-8), dict(dual='canonical'), dict(dual='floored', frame_floor=1e-3),
             dict(dual='additive', dual_eps=1e-6), dict(dual='additive', dual_eps=1e-12)]
    for rule, diag in itertools.product(rules, ('positive', 'mirrored')):
        rule = dict(rule, diagonal=diag)
        try:
            t, note = transform_for(dict(best, **rule))
            gotb = t.decode(c.astype(np.complex128))[0, 0]
        except (ValueError, AttributeError) as e:
            duals.append((np.inf, rule, str(e)))
            continue
        # the scale of decode() is a convention of its own (one-sided vs hermitian-extended spectrum): compare peak-normalised
        gn, wn = gotb / (np.abs(gotb).max() + 1e-30), wantb / (np.abs(wantb).max() + 1e-30)
        duals.append((float(np.abs(gn - wn).max()), rule, note))
    duals.sort(key=lambda r: r[0])
    print('dual-window rules ranked by max |oracle decode - cqt_pytorch decode| (both peak-normalised):')
    for err, rule, note in duals:
        print('  %.3e  %s %s' % (err, rule, note))
    return results, duals


if __name__ == '__main__':
    main()
    sys.exit(0)
