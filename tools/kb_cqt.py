"""Per-call timing of tt_cqt_forward / tt_cqt_inverse at the bench batch (KB_B clips x 3 s, default 64): HIP events on the caller's stream
around KB_N calls.  TTRAP_CQT_SPLIT selects the number of pipeline chunks (read once per process)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'timbre-trap_amd'))
from timbre_trap.framework import CQT                             # noqa: E402
from kb_wide import timeit                                       # noqa: E402

B, n = int(os.environ.get('KB_B', 64)), int(os.environ.get('KB_N', 50))
cqt = CQT(9, 60, 22050, 3).to('cuda')
a = torch.rand(B, 1, 66150, device='cuda') * 2 - 1
c = cqt(a)
by = B * 4688280
tf = timeit(lambda: cqt(a), n)
ti = timeit(lambda: cqt.decode(c), n)
print('split %s  B %d  forward %.4f ms  %.3f of HBM peak | inverse %.4f ms  %.3f of HBM peak' % (
    os.environ.get('TTRAP_CQT_SPLIT', 'default'), B, tf, by / tf / 1e6 / 8000.0, ti, by / ti / 1e6 / 8000.0))
