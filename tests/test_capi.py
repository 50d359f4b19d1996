"""CPU-only: the C-ABI library builds for gfx950, loads, and exports every symbol include/ttrap.h declares."""

import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'ttrap.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(tt_[a-z0-9_]+)\s*\(', text)))


def test_library_builds_and_exports_header_symbols():
    from timbre_trap import _hip
    path = _hip.build()
    assert os.path.exists(path)
    h = ctypes.CDLL(path)
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(h, name), 'symbol %s declared in include/ttrap.h is not exported' % name
    # the ctypes prototypes cover exactly the declared symbols
    assert sorted(_hip.EXPORTED_SYMBOLS) == declared
    lib = _hip.lib()
    assert lib.tt_arch() == b'gfx950'
    assert lib.tt_version() >= 1
    assert lib.tt_error_string(-1) == b'ttrap: bad argument'
    assert lib.tt_cqt_scratch_bytes(4, 540, 65649) > 4 * 33075 * 8 * 2


def test_no_cpu_fallback():
    """Device entry points refuse CPU tensors instead of silently computing somewhere else."""
    import pytest
    import torch
    from timbre_trap.framework import TimbreTrap, compute_reconstruction_loss
    model = TimbreTrap(22050, 9, 60, 3, model_complexity=1)
    with pytest.raises(RuntimeError):
        model.sliCQ(torch.zeros(1, 1, 66150))
    with pytest.raises(RuntimeError):
        model.encoder(torch.zeros(1, 2, 540, 4))
    with pytest.raises(RuntimeError):
        compute_reconstruction_loss(torch.zeros(1, 2, 540, 4), torch.zeros(1, 2, 540, 4))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'timbre-trap_amd')
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(base, f)).read()
                assert 'oracle' not in src.replace('# oracle', ''), '%s mentions the oracle' % f
