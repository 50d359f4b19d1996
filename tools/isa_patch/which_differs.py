"""Which outputs of tt_wide_rb_bwd differ between repeated identical calls (C = 8, dilation 1, bench height; TTRAP_LIB selects the build)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'timbre-trap_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
import test_gpu_determinism as td

for C, d in ((8, 1), (8, 2), (4, 1)):
    runs = td._block_backward(C, d, runs=8)
    names = ('dw1', 'db1', 'dw2', 'db2', 'dx')
    for name, idx in zip(names, range(5)):
        ref = runs[0][idx]
        bad = {}
        for r in runs[1:]:
            diff = (r[idx] != ref).flatten().nonzero().flatten().tolist()
            for i in diff:
                bad.setdefault(i, []).append(float((r[idx].flatten()[i] - ref.flatten()[i]).abs() / (ref.flatten()[i].abs() + 1e-30)))
        if bad and not (name == 'dw1'):
            print('C %d d %d %s: %d element(s) differ across 8 runs: %s' % (C, d, name, len(bad), {k: ['%.1e' % v for v in vs[:3]] for k, vs in list(bad.items())[:6]}))
    print('C %d d %d checked' % (C, d))
