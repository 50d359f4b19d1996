"""Per-launch timing of the residual-block kernels of ONE width / dilation at the bench shape (B 64, T 1024), for PMC passes:
KB_C (32; comma list), KB_D (1; comma list), KB_WHAT = comma list of fwd,fwdns,bwd,bwd1,bwdf,stride,stridefwd,edge, KB_N iterations (10)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'timbre-trap_amd'))
from timbre_trap import _hip                                     # noqa: E402
from timbre_trap._hip import check, ptr, stream_ptr              # noqa: E402
from kb_wide import timeit                                       # noqa: E402


def run_level(lib, st, C, dils, what, n):
    B, T = int(os.environ.get('KB_B', 64)), 1024
    H = {32: 65, 16: 133, 8: 269, 4: 540}[C]
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
    for d in dils:
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
            print('C%d d%d bwd    %.3f ms  %.2f TB/s (dy + x + dx)' % (C, d, t, npx * C * 6 / t / 1e9))
        if 'bwd1' in what and C >= 16:
            check(lib.tt_wide_rb_fwd(ptr(xb), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(yb), ptr(hb), B, C, H, T, d, st), 'fwd')
            t = timeit(lambda: check(lib.tt_wide_rb_bwd_onepass(ptr(xb), ptr(hb), ptr(gb), ptr(w1), ptr(w2), ptr(b2), ptr(dxb), ptr(dw1), ptr(db1),
                                                                ptr(dw2), ptr(db2), ptr(ws), B, C, H, T, d, st), 'bwd1'), n)
            print('C%d d%d bwd-onepass %.3f ms  %.2f TB/s (dy + x + dx)  %.2f TB/s (h1 + dy + x + dx)' % (C, d, t, npx * C * 6 / t / 1e9, npx * C * 8 / t / 1e9))
        if 'bwdf' in what and C >= 16:
            wsf = torch.empty(lib.tt_wide_fused_scratch_bytes(C), dtype=torch.uint8, device='cuda')
            t = timeit(lambda: check(lib.tt_wide_rb_bwd_fused(ptr(xb), ptr(gb), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(dxb), ptr(dw1),
                                                              ptr(db1), ptr(dw2), ptr(db2), ptr(wsf), B, C, H, T, d, st), 'bwdf'), n)
            print('C%d d%d bwd-fused %.3f ms  %.2f TB/s (x + dy + dx)' % (C, d, t, npx * C * 6 / t / 1e9))
    if 'stride' in what:
        from timbre_trap.framework import ops
        Ho = (H - 4) // 2 + 1
        x = ops.new_cl16(B, C, H, T, 'cuda').normal_()
        y = ops.new_cl16(B, 2 * C, Ho, T, 'cuda').normal_()
        dy = ops.new_cl16(B, 2 * C, Ho, T, 'cuda').normal_()
        dx = ops.new_cl16(B, C, H, T, 'cuda')
        w = torch.randn(2 * C, C, 4, 1, device='cuda') * 0.1
        dw, db, dbt = torch.zeros_like(w), torch.zeros(2 * C, device='cuda'), torch.zeros(C, device='cuda')
        wss = torch.empty(lib.tt_stride16_scratch_bytes(C), dtype=torch.uint8, device='cuda')
        p = H - (2 * Ho + 2)
        t = timeit(lambda: check(lib.tt_sconv16_bwd(ptr(x), ptr(y), ptr(dy), ptr(w), ptr(dx), ptr(dw), ptr(db), ptr(wss), B, C, H, T, st), 'b'), n)
        print('sconv C%-2d bwd %.3f ms' % (C, t))
        t = timeit(lambda: check(lib.tt_tconv16_bwd(ptr(y), ptr(x), ptr(dx), ptr(w), ptr(dy), ptr(dw), ptr(dbt), ptr(wss), B, C, Ho, T, p, st), 'b'), n)
        print('tconv C%-2d bwd %.3f ms' % (C, t))


def run_edge(lib, st, n):
    """Boundary 3x3 convolutions (2 <-> 4 channels) at the bench shape (H = 540)."""
    B, H, T = int(os.environ.get('KB_B', 64)), 540, 1024
    torch.manual_seed(0)
    x2 = torch.randn(B, 2, H, T, device='cuda')
    g2 = torch.randn(B, 2, H, T, device='cuda')
    x4 = torch.randn(B, H, T, 4, device='cuda').bfloat16()
    y4, g4 = torch.empty_like(x4), torch.randn(B, H, T, 4, device='cuda').bfloat16()
    w_in, b_in = torch.randn(4, 2, 3, 3, device='cuda') * 0.2, torch.randn(4, device='cuda') * 0.1
    w_out, b_out = torch.randn(2, 4, 3, 3, device='cuda') * 0.2, torch.randn(2, device='cuda') * 0.1
    dw_in, db_in, dw_out, db_out = torch.zeros_like(w_in), torch.zeros_like(b_in), torch.zeros_like(w_out), torch.zeros_like(b_out)
    ws = torch.empty(lib.tt_edge16_scratch_bytes(), dtype=torch.uint8, device='cuda')
    dx2, y2 = torch.empty_like(x2), torch.empty_like(x2)
    px = B * H * T
    t = timeit(lambda: check(lib.tt_convin16_fwd(ptr(x2), ptr(w_in), ptr(b_in), ptr(y4), B, H, T, st), 'cinf'), n)
    print('convin  fwd %.3f ms  %.2f TB/s (8 B in + 8 B out per pixel)' % (t, px * 16 / t / 1e9))
    t = timeit(lambda: check(lib.tt_convin16_bwd(ptr(x2), ptr(y4), ptr(g4), ptr(w_in), ptr(dx2), ptr(dw_in), ptr(db_in), ptr(ws), B, H, T, st), 'cinb'), n)
    print('convin  bwd %.3f ms  %.2f TB/s (x, y, dy, dx: 32 B per pixel)' % (t, px * 32 / t / 1e9))
    t = timeit(lambda: check(lib.tt_convout16_fwd(ptr(x4), ptr(w_out), ptr(b_out), ptr(y2), B, H, T, st), 'coutf'), n)
    print('convout fwd %.3f ms  %.2f TB/s (16 B per pixel)' % (t, px * 16 / t / 1e9))
    t = timeit(lambda: check(lib.tt_convout16_bwd(ptr(x4), ptr(g2), ptr(w_out), ptr(y4), ptr(dw_out), ptr(db_out), ptr(ws), B, H, T, st), 'coutb'), n)
    print('convout bwd %.3f ms  %.2f TB/s (x, dy, dx: 24 B per pixel)' % (t, px * 24 / t / 1e9))


def run_stride_fwd(lib, st, C, n):
    from timbre_trap.framework import ops
    B, T = int(os.environ.get('KB_B', 64)), 1024
    H = {32: 65, 16: 133, 8: 269, 4: 540}[C]
    Ho = (H - 4) // 2 + 1
    x = ops.new_cl16(B, C, H, T, 'cuda').normal_()
    y = ops.new_cl16(B, 2 * C, Ho, T, 'cuda').normal_()
    w = torch.randn(2 * C, C, 4, 1, device='cuda') * 0.1
    b2, b1 = torch.randn(2 * C, device='cuda') * 0.1, torch.randn(C, device='cuda') * 0.1
    gb = (x.numel() + y.numel()) * 2 / 1e9
    t = timeit(lambda: check(lib.tt_sconv16_fwd(ptr(x), ptr(w), ptr(b2), ptr(y), B, C, H, T, st), 'f'), n)
    print('sconv C%-2d fwd %.3f ms  %.2f TB/s' % (C, t, gb / t))
    t = timeit(lambda: check(lib.tt_tconv16_fwd(ptr(y), ptr(w), ptr(b1), ptr(x), B, C, Ho, T, H - (2 * Ho + 2), st), 'f'), n)
    print('tconv C%-2d fwd %.3f ms  %.2f TB/s' % (C, t, gb / t))


def main():
    lib, st = _hip.lib(), stream_ptr()
    what = os.environ.get('KB_WHAT', 'fwd,fwdns,bwd,bwdf').split(',')
    n = int(os.environ.get('KB_N', 10))
    dils = [int(v) for v in os.environ.get('KB_D', '1').split(',')]
    for C in [int(v) for v in os.environ.get('KB_C', '32').split(',')]:
        run_level(lib, st, C, dils, what, n)
        if 'stridefwd' in what:
            run_stride_fwd(lib, st, C, n)
    if 'edge' in what:
        run_edge(lib, st, n)


if __name__ == '__main__':
    main()
