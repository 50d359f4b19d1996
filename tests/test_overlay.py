"""
The package is an OVERLAY on a reference checkout: with ``timbre-trap_amd`` in front of the reference on
``sys.path``, the import block of reference experiments/train.py:1-9 and evaluate.py:1-3 must resolve -- framework from this
package, dataset wrappers / constants from the reference.  Third-party packages the image lacks are stubbed for import only.
Skipped where the reference is not present (e.g. the GPU box).
"""

import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = '/root/reference'

SCRIPT = textwrap.dedent('''
    import sys, types
    sys.path.insert(0, %r); sys.path.insert(1, %r)
    def stub(name, **attrs):
        m = types.ModuleType(name); m.__dict__.update(attrs); sys.modules[name] = m
    for n in ('torchaudio', 'librosa', 'mir_eval', 'jams', 'mido', 'sacred', 'torchmetrics'):
        try:
            __import__(n)
        except ImportError:
            stub(n)
    if not hasattr(sys.modules['sacred'], 'Experiment'):
        sys.modules['sacred'].Experiment = object
        stub('sacred.observers', FileStorageObserver=object)
    if 'torchmetrics.audio' not in sys.modules and not hasattr(sys.modules['torchmetrics'], '__path__'):
        stub('torchmetrics.audio', SignalDistortionRatio=object)
    # ---- reference experiments/train.py:1-9 ----
    from timbre_trap.datasets.MixedMultiPitch import URMP as URMP_Mixtures, Bach10 as Bach10_Mixtures, Su, TRIOS
    from timbre_trap.datasets.SoloMultiPitch import URMP as URMP_Stems, MedleyDB_Pitch, GuitarSet
    from timbre_trap.datasets.AudioMixtures import MedleyDB as MedleyDB_Mixtures, FMA
    from timbre_trap.datasets.AudioStems import MedleyDB as MedleyDB_Stems
    from timbre_trap.datasets import ComboDataset
    from timbre_trap.framework import *
    from timbre_trap.framework.objectives import *
    from timbre_trap.utils import *
    # ---- reference experiments/evaluate.py:1-3 ----
    from timbre_trap.datasets import NoteDataset
    import timbre_trap.framework.modules as fm, timbre_trap.utils as u, timbre_trap.datasets as d
    assert fm.__file__.startswith(%r), fm.__file__                      # the framework is OURS
    assert URMP_Mixtures.__module__.startswith('timbre_trap.datasets.MixedMultiPitch')
    assert sys.modules[URMP_Mixtures.__module__].__file__.startswith(%r)   # the dataset wrappers are the REFERENCE's
    assert constants.KEY_AUDIO and d.REFERENCE_DATASETS and u.REFERENCE_MODULES['data'].startswith(%r)
    for name in ('TimbreTrap', 'CQT', 'TimbreTrapMag', 'compute_reconstruction_loss', 'seed_everything', 'print_and_log', 'DataParallel',
                 'CosineWarmup', 'sum_gradient_norms', 'average_gradient_norms', 'get_max_gradient', 'get_max_gradient_norm',
                 'log_gradient_norms', 'MultipitchEvaluator', 'to_array', 'debug_nans', 'filter_non_peaks', 'threshold', 'constants'):
        assert name in globals(), name
    import torch
    assert issubclass(DataParallel, torch.nn.DataParallel)
    print('OVERLAY_OK')
''') % (os.path.join(ROOT, 'timbre-trap_amd'), REFERENCE, os.path.join(ROOT, 'timbre-trap_amd'), REFERENCE, REFERENCE)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, 'timbre_trap')), reason='no reference checkout on this machine')
def test_reference_script_imports_resolve_through_the_overlay():
    env = {k: v for k, v in os.environ.items() if k != 'PYTHONPATH'}
    out = subprocess.run([sys.executable, '-c', SCRIPT], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and 'OVERLAY_OK' in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]


def test_package_is_self_contained_without_a_reference():
    import timbre_trap.datasets as d
    import timbre_trap.utils as u
    assert hasattr(u, 'MultipitchEvaluator') and hasattr(u, 'FusedAdamW') and hasattr(d, 'PitchDataset')
