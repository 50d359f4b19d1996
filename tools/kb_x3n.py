"""Kernel-level timing of the narrow split-operand residual blocks (tt_x3n_rb_fwd) at the inference shape of BASELINE configs[1]
(96 chunks x 3 s): KB_C=4,8 KB_D=1,2,3 KB_N=10 KB_IO=pp,ps,ss,sp python tools/kb_x3n.py   (io: p = fp32 planar, s = split tensors)"""
import os
import sys
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'timbre-trap_amd'))
from timbre_trap import _hip
from timbre_trap._hip import check, ptr, stream_ptr
from kb_x3 import timeit

SHAPES = {4: (96, 540, 1024), 8: (96, 269, 1024)}


def main():
    lib, st = _hip.lib(), stream_ptr()
    n = int(os.environ.get('KB_N', 10))
    for C in [int(c) for c in os.environ.get('KB_C', '4,8').split(',')]:
        B, H, T = SHAPES[C]
        B = int(os.environ.get('KB_B', B))
        xp = torch.randn(B, C, H, T, device='cuda')
        xs = torch.randn(B, H, T, 2, C, device='cuda').half()
        yp, ys = torch.empty_like(xp), torch.empty_like(xs)
        w1 = torch.randn(C, C, 3, 3, device='cuda') / (3 * C ** 0.5)
        b1 = torch.randn(C, device='cuda') * 0.1
        w2 = torch.randn(C, C, 1, 1, device='cuda') / C ** 0.5
        b2 = torch.randn(C, device='cuda') * 0.1
        nbytes = 2 * B * C * H * T * 4
        for d in [int(v) for v in os.environ.get('KB_D', '1,2,3').split(',')]:
            for io in os.environ.get('KB_IO', 'ps,ss,sp').split(','):
                pin, pout = io[0] == 'p', io[1] == 'p'
                src, dst = (xp if pin else xs), (yp if pout else ys)
                t = timeit(lambda: check(lib.tt_x3n_rb_fwd(ptr(src), int(pin), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(dst), int(pout), B, C, H, T, d, st), 'x3n'), n)
                print('C%d d%d x3n %s  %.3f ms  %.2f TB/s (x + y at 4 bytes per element)' % (C, d, io, t, nbytes / t / 1e9))


if __name__ == '__main__':
    main()
