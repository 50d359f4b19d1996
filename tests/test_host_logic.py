"""
CPU-only tests of the host side: NSGT plan tables vs the oracle's, the pure-Python frame/time helpers
bit-exact vs values recorded from the reference, state_dict key compatibility, pickling / fork safety.
"""

import multiprocessing as mp
import pickle

import numpy as np
import torch

from oracle import autoencoder as oae
from oracle import nsgt
from timbre_trap.framework import CQT, TimbreTrap
from timbre_trap.framework import nsgt_plan


def test_plan_matches_oracle_tables():
    tab = nsgt.nsgt_tables(9, 60, 22050, 66150)
    plan = nsgt_plan.build_plan(9, 60, 22050, 66150)
    assert plan['M'] == tab['max_window_length'] == 1024 and plan['N'] == 66150
    for a, b in (('lengths', 'lengths'), ('positions', 'positions'), ('pad', 'pad'), ('start', 'start'),
                 ('win_off', 'win_off'), ('spec_index', 'spec_index')):
        assert np.array_equal(plan[a], tab[b]), a
    assert np.array_equal(plan['window'], tab['window'])
    assert np.array_equal(plan['dual'], tab['dual'])
    assert np.array_equal(plan['covered'], tab['covered'])
    # CSR gather is a permutation of the ragged positions grouped by spectral index
    off, idx = plan['gat_off'], plan['gat_idx']
    assert off[-1] == plan['sum_len'] and sorted(idx.tolist()) == list(range(plan['sum_len']))
    for j in (65, 94, 1000, 20000, 33000):
        assert all(plan['spec_index'][p] == j for p in idx[off[j]:off[j + 1]])
    bt = plan['bin_tab']
    assert bt.shape == (540, 4) and np.array_equal(bt[:, 0], tab['start'] + tab['pad'])
    # twiddles
    assert np.allclose(plan['twN'][1], [np.cos(2 * np.pi / 66150), -np.sin(2 * np.pi / 66150)])
    assert plan['twN'].shape == (33076, 2) and plan['twNc'].shape == (33075, 2)
    a = -2 * np.pi * 48 * 674 / 33075
    assert np.allclose(plan['twNc'][48 * 675 + 674], [np.cos(a), np.sin(a)])


def test_helpers_bit_exact(golden):
    g = golden('wrapper')
    cq = CQT(9, 60, 22050, 3)
    assert cq.block_length == 66150 and cq.max_window_length == 1024 and cq.n_bins == 540
    assert cq.sample_rate == 22050
    assert cq.hop_length == float(g['hop_length'])
    assert np.array_equal(cq.get_midi_freqs(), g['midi_freqs']) and cq.midi_freqs is cq.get_midi_freqs()
    assert [cq.get_expected_frames(int(n)) for n in g['frames_in']] == list(g['frames_out'])
    assert [cq.get_expected_samples(float(t)) for t in g['samples_in']] == list(g['samples_out'])
    assert np.array_equal(cq.get_times(3100), g['times_3100'])
    assert [cq.pad_to_block_length(torch.zeros(1, 1, int(n))).size(-1) for n in g['pad_lens_in']] == list(g['pad_lens_out'])
    c = torch.complex(torch.from_numpy(g['to_real_in_re']), torch.from_numpy(g['to_real_in_im']))
    r = CQT.to_real(c)
    assert np.array_equal(r.numpy(), g['to_real_out'])
    np.testing.assert_allclose(CQT.to_magnitude(r).numpy(), g['to_magnitude_out'], rtol=1e-6)
    cc = CQT.to_complex(r)
    assert np.array_equal(cc.real.numpy(), g['to_complex_re']) and np.array_equal(cc.imag.numpy(), g['to_complex_im'])
    m = torch.rand(2, 540, 7) * 3
    np.testing.assert_allclose(CQT.to_decibels(m).numpy(), nsgt.to_decibels(m.numpy()), rtol=1e-5, atol=1e-6)


def test_state_dict_keys_match_reference():
    for kw in (dict(latent_size=None, model_complexity=1, skip_connections=False),
               dict(latent_size=128, model_complexity=2, skip_connections=True)):
        model = TimbreTrap(22050, 9, 60, 3, **kw)
        shapes = oae.state_dict_shapes(540, **kw)
        sd = model.state_dict()
        assert list(sd.keys()) == list(shapes.keys())
        assert all(tuple(sd[k].shape) == shapes[k] for k in shapes)
        # reference checkpoints also carry cqt_pytorch buffers under sliCQ.*: tolerated on load
        extra = dict(oae.closed_form_state_dict(shapes))
        extra['sliCQ.windows'] = torch.zeros(3)
        extra['sliCQ.windows_range_indices'] = torch.zeros(3, dtype=torch.long)
        model.load_state_dict(extra, strict=True)
    n_params = sum(p.numel() for p in TimbreTrap(22050, 9, 60, 3, latent_size=128, model_complexity=2).parameters())
    assert n_params == 614490


def test_default_init_matches_torch_layers():
    """Same RNG consumption as the reference constructors: identical default weights under one seed."""
    torch.manual_seed(2)
    m = TimbreTrap(22050, 9, 60, 3, latent_size=128, model_complexity=2)
    torch.manual_seed(2)
    ref = torch.nn.Conv2d(2, 4, kernel_size=3, padding='same')
    assert torch.equal(m.encoder.convin[0].weight, ref.weight)


def _worker(blob, q):
    cq = pickle.loads(blob)
    q.put((cq.get_expected_frames(100000), float(cq.get_times(3)[2]), cq.hop_length))


def test_cqt_pickles_and_forks_without_touching_hip():
    cq = CQT(9, 60, 22050, 3)
    blob = pickle.dumps(cq)
    ctx = mp.get_context('fork')
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(blob, q))
    p.start()
    out = q.get(timeout=60)
    p.join()
    assert out == (1548, 0.005859375, 64.599609375)
    model = TimbreTrap(22050, 9, 60, 3)
    m2 = pickle.loads(pickle.dumps(model))
    assert m2.sliCQ.n_bins == 540


def test_gate_link_promises():
    """ops.GateLink (the forward-time handshake that lets a layer's backward take its gradient already multiplied by its ELU derivative):
    nothing is gated until a consumer promises; a promise that depends on another link's (the first transposed layer gates its dx only in
    its pregated form, i.e. when the level behind it gates in turn) follows that link at the time it is READ; ops.PREGATE = False hands
    out no links; a tap on an unlinked tensor is the tensor itself."""
    from timbre_trap.framework import ops
    link = ops.GateLink()
    assert not link.producer and not link.gated
    link.gated = True
    assert link.gated
    down = ops.GateLink()
    link.depends = down
    assert not link.gated                                    # the level behind has not promised
    down.gated = True
    assert link.gated
    down.gated = False
    assert not link.gated
    saved = ops.PREGATE
    try:
        ops.PREGATE = False
        assert ops.gate_link() is None
        ops.PREGATE = True
        assert isinstance(ops.gate_link(), ops.GateLink)
    finally:
        ops.PREGATE = saved
    y = torch.zeros(1, 4, 3, 8, requires_grad=True)
    assert ops.gate_tap(y, None) is y and ops.gate_tap(y, ops.GateLink()) is y      # no producer marked: nothing to do


def test_bench_core_cap_applies_before_torch_is_imported():
    """`bench.py --cores K` (round 6: the one-GPU proxy for K host cores per rank) restricts the process's affinity when the module is imported --
    before torch and its thread pools -- and tells OpenMP the same number."""
    import os
    import subprocess
    import sys
    import pytest
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if len(os.sched_getaffinity(0)) < 2:
        pytest.skip('needs two cores to cap to one fewer')
    code = ("import sys, os; sys.argv = ['bench.py', '--cores', '1']; sys.path.insert(0, %r); import bench; "
            "print(len(os.sched_getaffinity(0)), bench.CORE_CAP, os.environ.get('OMP_NUM_THREADS'), bench.available_cores())" % root)
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split() == ['1', '1', '1', '1'], out.stdout


def test_default_dual_window_can_be_pinned_from_the_environment():
    """TTRAP_CQT_DUAL (round-5 advisor finding): the reference's scripts build ``TimbreTrap(...)`` and never see ``conventions=``; the environment pins
    the dual window they get -- additive (default) | canonical | floored -- and anything else is refused at import."""
    import os
    import subprocess
    import sys
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'timbre-trap_amd')
    code = "import sys; sys.path.insert(0, %r); from timbre_trap.framework import nsgt_plan as p; print(p.DEFAULT_CONVENTIONS.dual)" % pkg
    for env, want in ((None, 'additive'), ('floored', 'floored'), ('canonical', 'canonical')):
        e = dict(os.environ)
        e.pop('TTRAP_CQT_DUAL', None)
        if env:
            e['TTRAP_CQT_DUAL'] = env
        out = subprocess.run([sys.executable, '-c', code], env=e, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and out.stdout.strip() == want, (env, out.stdout, out.stderr[-500:])
    bad = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, TTRAP_CQT_DUAL='flored'), capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and 'TTRAP_CQT_DUAL' in bad.stderr
