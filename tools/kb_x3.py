"""Kernel-level timing of the split-operand (x3) residual blocks next to the fp32 and bf16 kernels (bench shape, 64 clips).
   KB_C=16,32 KB_D=1,2,3 KB_N=20 python tools/kb_x3.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'timbre-trap_amd'))
from timbre_trap import _hip
from timbre_trap._hip import check, ptr, stream_ptr
from timbre_trap.framework import ops

SHAPES = {32: (64, 65, 1024), 16: (64, 133, 1024)}


def timeit(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    lib, st = _hip.lib(), stream_ptr()
    n = int(os.environ.get('KB_N', 20))
    for C in [int(c) for c in os.environ.get('KB_C', '16,32').split(',')]:
        B, H, T = SHAPES[C]
        B = int(os.environ.get('KB_B', B))
        x = torch.randn(B, C, H, T, device='cuda')
        w1 = torch.randn(C, C, 3, 3, device='cuda') / (3 * C ** 0.5)
        b1 = torch.randn(C, device='cuda') * 0.1
        w2 = torch.randn(C, C, 1, 1, device='cuda') / C ** 0.5
        b2 = torch.randn(C, device='cuda') * 0.1
        a = torch.empty((B, H, T, 2, C), dtype=torch.float16, device='cuda')
        b = torch.empty_like(a)
        y = torch.empty_like(x)
        gb = x.numel() * 4 / 1e9
        ms = timeit(lambda: check(lib.tt_x3_pack(ptr(x), ptr(a), B, C, H, T, st), 'pack'), n)
        print('C%d pack    %.3f ms  %.2f TB/s (read + write)' % (C, ms, 2 * gb / ms))
        ms = timeit(lambda: check(lib.tt_x3_unpack(ptr(a), ptr(y), B, C, H, T, st), 'unpack'), n)
        print('C%d unpack  %.3f ms  %.2f TB/s (read + write)' % (C, ms, 2 * gb / ms))
        for d in [int(v) for v in os.environ.get('KB_D', '1,2,3').split(',')]:
            ms = timeit(lambda: check(lib.tt_x3_rb_fwd(ptr(a), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(b), 0, B, C, H, T, d, st), 'x3'), n)
            msp = timeit(lambda: check(lib.tt_x3_rb_fwd(ptr(a), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(y), 1, B, C, H, T, d, st), 'x3'), n)
            with torch.no_grad():
                ops.X3_INFER = False
                ms32 = timeit(lambda: ops.ResBlockFn.apply(x, w1, b1, w2, b2, d), n)
            xb = torch.empty((B, H, T, C), dtype=torch.bfloat16, device='cuda')
            yb = torch.empty_like(xb)
            ms16 = timeit(lambda: check(lib.tt_wide_rb_fwd(ptr(xb), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(yb), None, B, C, H, T, d, st), 'bf16'), n)
            print('C%d d%d x3 %.3f ms  %.2f TB/s (x + y, 4 bytes per element) | planar out %.3f ms | fp32 kernel %.3f ms | bf16 kernel %.3f ms' % (C, d, ms, 2 * gb / ms, msp, ms32, ms16))

        # strided layers on x3 tensors next to the fp32 kernels
        wS = torch.randn(2 * C, C, 4, 1, device='cuda') / (2 * C ** 0.5)
        bS = torch.randn(2 * C, device='cuda') * 0.1
        Ho = (H - 4) // 2 + 1
        y3 = torch.empty((B, Ho, T, 2, 2 * C), dtype=torch.float16, device='cuda')
        yp = torch.empty((B, 2 * C, Ho, T), device='cuda')
        ms3 = timeit(lambda: check(lib.tt_x3_sconv_fwd(ptr(a), 0, ptr(wS), ptr(bS), ptr(y3), 0, B, C, H, T, st), 's'), n)
        msp = timeit(lambda: check(lib.tt_x3_sconv_fwd(ptr(a), 0, ptr(wS), ptr(bS), ptr(yp), 1, B, C, H, T, st), 's'), n)
        with torch.no_grad():
            ms32 = timeit(lambda: ops.StridedConvFn.apply(x, wS, bS), n)
        gbs = (x.numel() + yp.numel()) * 4 / 1e9
        print('C%d sconv x3 -> x3 %.3f ms %.2f TB/s | x3 -> planar %.3f ms | fp32 kernel %.3f ms' % (C, ms3, gbs / ms3, msp, ms32))
        if C == 32:
            wT = torch.randn(32, 16, 4, 1, device='cuda') / 8
            bT = torch.randn(16, device='cuda') * 0.1
            Ht = 2 * H + 3
            z3 = torch.empty((B, Ht, T, 2, 16), dtype=torch.float16, device='cuda')
            mst = timeit(lambda: check(lib.tt_x3_tconv_fwd(ptr(a), 0, ptr(wT), ptr(bT), ptr(z3), 0, B, 16, H, T, 1, st), 't'), n)
            with torch.no_grad():
                mst32 = timeit(lambda: ops.TransposedConvFn.apply(x, wT, bT, 1), n)
            gbt = (x.numel() + z3.numel() // 2) * 4 / 1e9
            print('C32 -> 16 tconv x3 -> x3 %.3f ms %.2f TB/s | fp32 kernel %.3f ms' % (mst, gbt / mst, mst32))

    # the two layers that ENTER the split-operand part from fp32 planar tensors
    B = int(os.environ.get('KB_B', 64))
    x8 = torch.randn(B, 8, 269, 1024, device='cuda')
    w8 = torch.randn(16, 8, 4, 1, device='cuda') / 4
    b8 = torch.randn(16, device='cuda') * 0.1
    y16 = torch.empty((B, 133, 1024, 2, 16), dtype=torch.float16, device='cuda')
    ms = timeit(lambda: check(lib.tt_x3_sconv_fwd(ptr(x8), 1, ptr(w8), ptr(b8), ptr(y16), 0, B, 8, 269, 1024, st), 's8'), n)
    with torch.no_grad():
        ms32 = timeit(lambda: ops.StridedConvFn.apply(x8, w8, b8), n)
    print('C8 -> 16 sconv planar -> x3 %.3f ms %.2f TB/s | fp32 kernel %.3f ms (+ pack of its output)' % (ms, (x8.numel() * 4 + y16.numel() * 2) / 1e9 / ms, ms32))
    x64 = torch.randn(B, 64, 31, 1024, device='cuda')
    w64 = torch.randn(64, 32, 4, 1, device='cuda') / 11
    b32 = torch.randn(32, device='cuda') * 0.1
    y32 = torch.empty((B, 65, 1024, 2, 32), dtype=torch.float16, device='cuda')
    ms = timeit(lambda: check(lib.tt_x3_tconv_fwd(ptr(x64), 1, ptr(w64), ptr(b32), ptr(y32), 0, B, 32, 31, 1024, 1, st), 't64'), n)
    with torch.no_grad():
        ms32 = timeit(lambda: ops.TransposedConvFn.apply(x64, w64, b32, 1), n)
    print('C64 -> 32 tconv planar -> x3 %.3f ms %.2f TB/s | fp32 kernel %.3f ms (+ pack of its output)' % (ms, (x64.numel() * 4 + y32.numel() * 2) / 1e9 / ms, ms32))


if __name__ == '__main__':
    main()
