"""
Comparison with fixtures recorded from the real ``cqt_pytorch`` (tools/pin_cqt.py).  The package is absent from this image, so
normally no fixture exists and these tests SKIP with that reason -- the CQT is then parity-unpinned (oracle/nsgt.py).  The
day tests/golden/cqt_pytorch_pin.npz exists, the float64 oracle (CPU) and the HIP transform (GPU) are held to it.
"""

import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PIN = os.path.join(ROOT, 'tests', 'golden', 'cqt_pytorch_pin.npz')
needs_pin = pytest.mark.skipif(not os.path.exists(PIN), reason='no cqt_pytorch fixture (package not installable here): CQT parity unpinned')


def test_pin_tool_is_a_clean_no_op_without_cqt_pytorch():
    try:
        import cqt_pytorch  # noqa: F401
        pytest.skip('cqt_pytorch is importable: run tools/pin_cqt.py to record the fixture')
    except ImportError:
        pass
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'pin_cqt.py')], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and 'parity-unpinned' in out.stdout and not os.path.exists(PIN)


@needs_pin
def test_oracle_matches_cqt_pytorch_fixture():
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import pin_cqt
    from oracle.nsgt_dense import DenseNSGT
    g = np.load(PIN)
    x = pin_cqt.pin_audio().astype(np.float64)
    got = DenseNSGT(9, 60, 22050, 66150).encode(x)[0, 0, ::pin_cqt.BIN_STRIDE, ::pin_cqt.FRAME_STRIDE]
    assert np.abs(got - g['coeff_sub']).max() <= 1e-4 * float(g['coeff_absmax'])


@needs_pin
@pytest.mark.gpu
def test_hip_cqt_matches_cqt_pytorch_fixture():
    import torch
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import pin_cqt
    from timbre_trap.framework import CQT
    g = np.load(PIN)
    cq = CQT(9, 60, 22050, 3).cuda()
    c = cq.encode(torch.from_numpy(pin_cqt.pin_audio()).cuda()).cpu().numpy()
    got = c[0, 0, ::pin_cqt.BIN_STRIDE, ::pin_cqt.FRAME_STRIDE]
    assert np.abs(got - g['coeff_sub']).max() <= 1e-4 * float(g['coeff_absmax'])        # north_star: 1e-4 rel on coefficients
