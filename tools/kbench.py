"""
Micro-benchmark of single C-ABI entry points at the bench shapes (B=64, T=1024), timed with HIP events on the
launch stream.  Usage:  python tools/kbench.py [op ...]     ops: rb_fwd rb_bwd sconv tconv cqt small all
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'timbre-trap_amd'))

import torch  # noqa: E402
from timbre_trap.framework import ops  # noqa: E402

LEVELS = {4: 540, 8: 269, 16: 133, 32: 65}


def timeit(fn, iters=int(os.environ.get('KB_ITERS', 5)), warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    which = sys.argv[1:] or ['all']
    B, T = int(os.environ.get('KB_B', 64)), 1024
    dev = 'cuda'
    for C, H in LEVELS.items():
        if os.environ.get('KB_C') and int(os.environ['KB_C']) != C:
            continue
        x = torch.randn(B, C, H, T, device=dev)
        px = B * H * T
        for d in (1, 2, 3):
            w1 = torch.randn(C, C, 3, 3, device=dev) * 0.1
            b1 = torch.randn(C, device=dev) * 0.1
            w2 = torch.randn(C, C, 1, 1, device=dev) * 0.1
            b2 = torch.randn(C, device=dev) * 0.1
            if 'rb_fwd' in which or 'all' in which:
                xg = x.clone().requires_grad_(True)          # training forward: the hidden activation is written too
                ms = timeit(lambda: ops.ResBlockFn.apply(xg, w1, b1, w2, b2, d))
                fl = 2.0 * 10 * C * C * px
                print('rb_fwd  C=%2d d=%d  %7.3f ms  %6.1f TFLOP/s  %6.0f GB/s(alg)' % (C, d, ms, fl / ms / 1e9, 2 * 4 * C * px / ms / 1e6))
            if 'rb_bwd' in which or 'all' in which:
                xr = x.clone().requires_grad_(True)
                ws_ = [t.clone().requires_grad_(True) for t in (w1, b1, w2, b2)]
                y = ops.ResBlockFn.apply(xr, *ws_, d)
                gy = torch.randn_like(y)
                ms = timeit(lambda: torch.autograd.grad(y, [xr] + ws_, gy, retain_graph=True))
                fl = 2.0 * (9 + 1 + 1 + 1 + 9 + 9) * C * C * px
                print('rb_bwd  C=%2d d=%d  %7.3f ms  %6.1f TFLOP/s' % (C, d, ms, fl / ms / 1e9))
        if 'sconv' in which or 'all' in which:
            w = torch.randn(2 * C, C, 4, 1, device=dev) * 0.1
            b = torch.randn(2 * C, device=dev) * 0.1
            ms = timeit(lambda: ops.StridedConvFn.apply(x, w, b))
            xr = x.clone().requires_grad_(True)
            wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
            y = ops.StridedConvFn.apply(xr, wr, br)
            gy = torch.randn_like(y)
            msb = timeit(lambda: torch.autograd.grad(y, [xr, wr, br], gy, retain_graph=True))
            print('sconv   C=%2d      fwd %7.3f ms  bwd %7.3f ms' % (C, ms, msb))
        if 'tconv' in which or 'all' in which:
            Hh = (H - 4) // 2 + 1
            x2 = torch.randn(B, 2 * C, Hh, T, device=dev)
            w = torch.randn(2 * C, C, 4, 1, device=dev) * 0.1
            b = torch.randn(C, device=dev) * 0.1
            ms = timeit(lambda: ops.TransposedConvFn.apply(x2, w, b, H % 2))
            xr = x2.clone().requires_grad_(True)
            wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
            y = ops.TransposedConvFn.apply(xr, wr, br, H % 2)
            gy = torch.randn_like(y)
            msb = timeit(lambda: torch.autograd.grad(y, [xr, wr, br], gy, retain_graph=True))
            print('tconv   C=%2d      fwd %7.3f ms  bwd %7.3f ms' % (C, ms, msb))
    if 'latent' in which or 'all' in which:
        x = torch.randn(B, 64, 31, T, device=dev, requires_grad=True)
        w = (torch.randn(128, 64, 31, 1, device=dev) * 0.02).requires_grad_(True)
        b = torch.zeros(128, device=dev, requires_grad=True)
        ms = timeit(lambda: ops.LatentEncodeFn.apply(x, w, b))
        y = ops.LatentEncodeFn.apply(x, w, b)
        gy = torch.randn_like(y)
        msx = timeit(lambda: torch.autograd.grad(y, [x], gy, retain_graph=True))
        msw = timeit(lambda: torch.autograd.grad(y, [w, b], gy, retain_graph=True))
        fl = 2.0 * 128 * 1984 * T * B
        print('convlat fwd %6.3f ms (%5.1f TF)  dgrad %6.3f ms  wgrad+bias %6.3f ms' % (ms, fl / ms / 1e9, msx, msw))
        z = torch.randn(B, 129, T, device=dev, requires_grad=True)
        w = (torch.randn(129, 64, 31, 1, device=dev) * 0.02).requires_grad_(True)
        b = torch.zeros(64, device=dev, requires_grad=True)
        ms = timeit(lambda: ops.LatentDecodeFn.apply(z, w, b))
        y = ops.LatentDecodeFn.apply(z, w, b)
        gy = torch.randn_like(y)
        msx = timeit(lambda: torch.autograd.grad(y, [z], gy, retain_graph=True))
        msw = timeit(lambda: torch.autograd.grad(y, [w, b], gy, retain_graph=True))
        print('dec convin fwd %6.3f ms  dgrad(+gate) %6.3f ms  wgrad+bias(+gate) %6.3f ms' % (ms, msx, msw))
    if 'cqt' in which or 'all' in which:
        from timbre_trap.framework import CQT
        cq = CQT(9, 60, 22050, 3).to(dev)
        a = torch.rand(B, 1, 66150, device=dev) * 2 - 1
        ms = timeit(lambda: cq(a), iters=20)
        print('cqt fwd B=%d  %7.3f ms  %6.0f GB/s (alg 4,688,280 B/clip)' % (B, ms, B * 4688280 / ms / 1e6))
        c = cq(a)
        ms = timeit(lambda: cq.decode(c), iters=20)
        print('cqt inv B=%d  %7.3f ms  %6.0f GB/s' % (B, ms, B * 4688280 / ms / 1e6))


if __name__ == '__main__':
    main()
