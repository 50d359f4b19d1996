"""Per-launch timing of the residual-block kernels of ONE width / dilation at the bench shape (B 64, T 1024), for PMC passes:
KB_C (32), KB_D (1), KB_WHAT = comma list of fwd,fwdns,bwd,bwdf,lvl (default all), KB_N iterations (10)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'timbre-trap_amd'))
from timbre_trap import _hip                                     # noqa: E402
from timbre_trap._hip import check, ptr, stream_ptr              # noqa: E402
from kb_wide import timeit                                       # noqa: E402


def main():
    lib, st = _hip.lib(), stream_ptr()
    C, d = int(os.environ.get('KB_C', 32)), int(os.environ.get('KB_D', 1))
    B, T = int(os.environ.get('KB_B', 64)), 1024
    H = {32: 65, 16: 133, 8: 269, 4: 540}[C]
    what = os.environ.get('KB_WHAT', 'fwd,fwdns,bwd,bwdf').split(',')
    n = int(os.environ.get('KB_N', 10))
    torch.manual_seed(0)
    w1 = torch.randn(C, C, 3, 3, device='cuda') * 0.05
    w2 = torch.randn(C, C, 1, 1, device='cuda') * 0.1
    b1 = torch.randn(C, device='cuda') * 0.1
    b2 = torch.randn(C, device='cuda') * 0.1
    xb = torch.randn(B, H, T, C, device='cuda').bfloat16()
    gb = torch.randn(B, H, T, C, device='cuda').bfloat16()
    yb, hb, dxb = (torch.empty_like(xb) for _ in range(3))
    ws = torch.empty(lib.tt_wide_scratch_bytes(B, C, H, T), dtype=torch.uint8, device='cuda')
    dw1, dw2, db1, db2 = torch.zeros_like(w1), torch.zeros_like(w2), torch.zeros_like(b1), torch.zeros_like(b2)
    npx = B * H * T
    if 'fwd' in what:
        t = timeit(lambda: check(lib.tt_wide_rb_fwd(ptr(xb), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(yb), ptr(hb), B, C, H, T, d, st), 'fwd'), n)
        print('C%d d%d fwd    %.3f ms  %.2f TB/s (x + y + h1)' % (C, d, t, npx * C * 6 / t / 1e9))
    if 'fwdns' in what:
        t = timeit(lambda: check(lib.tt_wide_rb_fwd(ptr(xb), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(yb), None, B, C, H, T, d, st), 'fwd'), n)
        print('C%d d%d fwd-ns %.3f ms  %.2f TB/s (x + y)' % (C, d, t, npx * C * 4 / t / 1e9))
    if 'bwd' in what:
        check(lib.tt_wide_rb_fwd(ptr(xb), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(yb), ptr(hb), B, C, H, T, d, st), 'fwd')
        t = timeit(lambda: check(lib.tt_wide_rb_bwd(ptr(xb), ptr(hb), ptr(gb), ptr(w1), ptr(w2), ptr(b2), ptr(dxb), ptr(dw1), ptr(db1),
                                                    ptr(dw2), ptr(db2), ptr(ws), B, C, H, T, d, st), 'bwd'), n)
        print('C%d d%d bwd    %.3f ms' % (C, d, t))
    if 'bwdf' in what and C >= 16:
        wsf = torch.empty(lib.tt_wide_fused_scratch_bytes(C), dtype=torch.uint8, device='cuda')
        t = timeit(lambda: check(lib.tt_wide_rb_bwd_fused(ptr(xb), ptr(gb), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(dxb), ptr(dw1),
                                                          ptr(db1), ptr(dw2), ptr(db2), ptr(wsf), B, C, H, T, d, st), 'bwdf'), n)
        print('C%d d%d bwd-fused %.3f ms  %.2f TB/s (x + dy + dx)' % (C, d, t, npx * C * 6 / t / 1e9))


if __name__ == '__main__':
    main()
