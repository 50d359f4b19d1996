"""Timing of the bf16 channels-last strided / transposed layers at the bench shapes (B 64, T 1024): ms per call."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'timbre-trap_amd'))
sys.path.insert(0, ROOT)
from timbre_trap import _hip                                     # noqa: E402
from timbre_trap._hip import check, ptr, stream_ptr              # noqa: E402
from timbre_trap.framework import ops                            # noqa: E402
from tools.kb_wide import timeit                                 # noqa: E402


def main():
    _hip.build()
    lib, st = _hip.lib(), stream_ptr()
    B, T = int(os.environ.get('KB_B', 64)), 1024
    for C, H in ((4, 540), (8, 269), (16, 133), (32, 65)):
        Ho = (H - 4) // 2 + 1
        x = ops.new_cl16(B, C, H, T, 'cuda').normal_()
        y = ops.new_cl16(B, 2 * C, Ho, T, 'cuda')
        dy = ops.new_cl16(B, 2 * C, Ho, T, 'cuda').normal_()
        dx = ops.new_cl16(B, C, H, T, 'cuda')
        w = torch.randn(2 * C, C, 4, 1, device='cuda') * 0.1
        b = torch.randn(2 * C, device='cuda') * 0.1
        dw, db = torch.zeros_like(w), torch.zeros_like(b)
        ws = torch.empty(lib.tt_stride16_scratch_bytes(C), dtype=torch.uint8, device='cuda')
        gb = (x.numel() + y.numel()) * 2 / 1e9
        t = timeit(lambda: check(lib.tt_sconv16_fwd(ptr(x), ptr(w), ptr(b), ptr(y), B, C, H, T, st), 'f'))
        print('sconv C%-2d fwd %.3f ms  %.2f TB/s' % (C, t, gb / t))
        t = timeit(lambda: check(lib.tt_sconv16_bwd(ptr(x), ptr(y), ptr(dy), ptr(w), ptr(dx), ptr(dw), ptr(db), ptr(ws), B, C, H, T, st), 'b'))
        print('sconv C%-2d bwd %.3f ms' % (C, t))
        # transposed: 2C x Ho rows -> C x (2 Ho + 2 + p) rows
        p = H - (2 * Ho + 2)
        bt = torch.randn(C, device='cuda') * 0.1
        dbt = torch.zeros_like(bt)
        t = timeit(lambda: check(lib.tt_tconv16_fwd(ptr(y), ptr(w), ptr(bt), ptr(x), B, C, Ho, T, p, st), 'f'))
        print('tconv C%-2d fwd %.3f ms  %.2f TB/s' % (C, t, gb / t))
        t = timeit(lambda: check(lib.tt_tconv16_bwd(ptr(y), ptr(x), ptr(dx), ptr(w), ptr(dy), ptr(dw), ptr(dbt), ptr(ws), B, C, Ho, T, p, st), 'b'))
        print('tconv C%-2d bwd %.3f ms' % (C, t))
        # round 5: the same two backward passes from a gradient that arrives already gated (no read of the saved output)
        t = timeit(lambda: check(lib.tt_sconv16_bwd_pregated(ptr(x), ptr(dy), ptr(w), ptr(dx), ptr(dw), ptr(db), ptr(ws), B, C, H, T, st), 'b'))
        print('sconv C%-2d bwd pregated %.3f ms' % (C, t))
        t = timeit(lambda: check(lib.tt_tconv16_bwd_pregated(ptr(y), ptr(dx), ptr(w), ptr(dy), ptr(dw), ptr(dbt), ptr(ws), B, C, Ho, T, p,
                                                              1 if C in (16, 32) else 0, st), 'b'))
        print('tconv C%-2d bwd pregated %.3f ms' % (C, t))


if __name__ == '__main__':
    main()
