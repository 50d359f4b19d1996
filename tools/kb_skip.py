"""Per-launch timing of the skip-join kernels (round 6) at the bench shape (B 64 clips per embedding, pair decode: 2 B in the decoder, T 1024), for the
PMC passes of tools/r06_profile.sh: tt_skip_join16_fwd / _bwd (plain, gated, accumulating) at every level and tt_wide_rb_fwd_join against
tt_wide_rb_fwd at dilation 3 (the last block of a decoder level).  KB_N iterations (10), KB_C comma list (4,8,16,32,64)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'timbre-trap_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from timbre_trap import _hip                                     # noqa: E402
from timbre_trap._hip import check, ptr, stream_ptr              # noqa: E402
from kb_wide import timeit                                       # noqa: E402


def main():
    lib, st = _hip.lib(), stream_ptr()
    n_it = int(os.environ.get('KB_N', 10))
    B, T = int(os.environ.get('KB_B', 64)), 1024
    w = torch.tensor([0.7, 1.1, 0.9, 1.3, 0.8], device='cuda')
    for C in [int(c) for c in os.environ.get('KB_C', '4,8,16,32,64').split(',')]:
        H = {64: 31, 32: 65, 16: 133, 8: 269, 4: 540}[C]
        e = torch.randn(B, H, T, C, device='cuda').bfloat16()
        y, g = (torch.randn(2 * B, H, T, C, device='cuda').bfloat16() for _ in range(2))
        out, de = torch.empty_like(y), torch.empty_like(e)
        ds = torch.zeros(5, device='cuda')
        n = e.numel()
        unit = n * 2 / 1e9                                        # GB of one embedding-sized tensor
        t = timeit(lambda: check(lib.tt_skip_join16_fwd(ptr(y), ptr(e), ptr(w), 2, ptr(out), n, 2, st), 'f'), n_it)
        print('C%-2d skip_join_fwd  reps 2        %.3f ms  %.2f TB/s (2 y + e in, 2 out)' % (C, t, 5 * unit / t))
        for flags, name in ((0, 'plain'), (1, 'gated'), (3, 'gated, accumulating')):
            t = timeit(lambda: check(lib.tt_skip_join16_bwd(ptr(g), ptr(e), ptr(w), 2, ptr(de), ptr(ds), n, 2, flags, st), 'b'), n_it)
            units = 5 if flags & 2 else 4
            print('C%-2d skip_join_bwd  reps 2 %-20s %.3f ms  %.2f TB/s (%d tensor units)' % (C, name, t, units * unit / t, units))
        if C == 64:
            continue
        w1, w2 = torch.randn(C, C, 3, 3, device='cuda') * 0.05, torch.randn(C, C, 1, 1, device='cuda') * 0.1
        b1, b2 = torch.randn(C, device='cuda') * 0.1, torch.randn(C, device='cuda') * 0.1
        h1 = torch.empty_like(y)
        t0 = timeit(lambda: check(lib.tt_wide_rb_fwd(ptr(y), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(out), ptr(h1), 2 * B, C, H, T, 3, st), 'f'), n_it)
        t1 = timeit(lambda: check(lib.tt_wide_rb_fwd_join(ptr(y), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(out), ptr(h1), ptr(e), ptr(w), 2, B, 2 * B, C, H, T, 3, st), 'f'), n_it)
        print('C%-2d block forward d3 over 2 B clips: %.3f ms, with the join in its epilogue %.3f ms (+%.0f us for one more embedding read)' % (C, t0, t1, 1e3 * (t1 - t0)))


if __name__ == '__main__':
    main()
