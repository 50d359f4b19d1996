"""Timing of the bf16-storage wide-level kernels at the bench shapes (B 64, T 1024): ms per launch by HIP events."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'timbre-trap_amd'))
from timbre_trap import _hip                                     # noqa: E402
from timbre_trap._hip import check, ptr, stream_ptr              # noqa: E402


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    _hip.build()
    lib, st = _hip.lib(), stream_ptr()
    B, T = int(os.environ.get('KB_B', 64)), 1024
    only = os.environ.get('KB_ONLY')
    for C, H in ((32, 65), (16, 133), (8, 269), (4, 540)):
        if only and int(only) != C:
            continue
        x = torch.randn(B, C, H, T, device='cuda')
        w1 = torch.randn(C, C, 3, 3, device='cuda') * 0.05
        w2 = torch.randn(C, C, 1, 1, device='cuda') * 0.1
        b1 = torch.randn(C, device='cuda') * 0.1
        b2 = torch.randn(C, device='cuda') * 0.1
        xb = torch.empty((B, H, T, C), dtype=torch.bfloat16, device='cuda')
        yb, hb, gb, dxb = (torch.empty_like(xb) for _ in range(4))
        ws = torch.empty(lib.tt_wide_scratch_bytes(B, C, H, T), dtype=torch.uint8, device='cuda')
        dw1, dw2, db1, db2 = torch.zeros_like(w1), torch.zeros_like(w2), torch.zeros_like(b1), torch.zeros_like(b2)
        npx = B * H * T
        t = timeit(lambda: check(lib.tt_wide_pack(ptr(x), ptr(xb), B, C, H, T, st), 'pack'))
        print('C%d pack      %.3f ms  %.2f TB/s' % (C, t, npx * C * 6 / t / 1e9))
        check(lib.tt_wide_pack(ptr(x), ptr(gb), B, C, H, T, st), 'pack')
        t = timeit(lambda: check(lib.tt_wide_unpack(ptr(xb), ptr(x), B, C, H, T, st), 'unpack'))
        print('C%d unpack    %.3f ms  %.2f TB/s' % (C, t, npx * C * 6 / t / 1e9))
        for d in (1, 2, 3):
            t = timeit(lambda: check(lib.tt_wide_rb_fwd(ptr(xb), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(yb), ptr(hb), B, C, H, T, d, st), 'fwd'))
            print('C%d d%d fwd    %.3f ms  %.2f TB/s (x + y + h1)' % (C, d, t, npx * C * 6 / t / 1e9))
            t = timeit(lambda: check(lib.tt_wide_rb_fwd(ptr(xb), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(yb), None, B, C, H, T, d, st), 'fwd'))
            print('C%d d%d fwd-ns %.3f ms' % (C, d, t))
            t = timeit(lambda: check(lib.tt_wide_rb_bwd(ptr(xb), ptr(hb), ptr(gb), ptr(w1), ptr(w2), ptr(b2), ptr(dxb), ptr(dw1), ptr(db1),
                                                        ptr(dw2), ptr(db2), ptr(ws), B, C, H, T, d, st), 'bwd'))
            print('C%d d%d bwd    %.3f ms' % (C, d, t))
            if C >= 16:
                wsf = torch.empty(lib.tt_wide_fused_scratch_bytes(C), dtype=torch.uint8, device='cuda')
                t = timeit(lambda: check(lib.tt_wide_rb_bwd_fused(ptr(xb), ptr(gb), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(dxb), ptr(dw1),
                                                                  ptr(db1), ptr(dw2), ptr(db2), ptr(wsf), B, C, H, T, d, st), 'bwdf'))
                print('C%d d%d bwd-fused %.3f ms  %.2f TB/s (x + dy + dx)  [tile set %s, per CU %s]' % (
                    C, d, t, npx * C * 6 / t / 1e9, os.environ.get('TTRAP_FBWD_TILE', '0'), os.environ.get('TTRAP_FBWD_PER_CU', '2')))


if __name__ == '__main__':
    main()
