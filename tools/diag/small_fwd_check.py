"""Outputs (y, h1) of tt_resblock_fwd at C = 4, 8 for a set of shapes -> a .pt file; run once per switch setting
   (TTRAP_SMALL_FWD4=0 / 1) and compare:  python tools/diag/small_fwd_check.py cmp a.pt b.pt   (bit equality expected).
   `time` as first argument: ms per launch at the inference planes (B 96)."""
import os
import sys
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'timbre-trap_amd'))


def call(lib, x, w1, b1, w2, b2, y, h1, d):
    from timbre_trap._hip import check, ptr, stream_ptr
    B, C, H, T = x.shape
    check(lib.tt_resblock_fwd(ptr(x), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(y), ptr(h1), B, C, H, T, d, 0, stream_ptr()), 'fwd')


def params(C, g):
    w1 = ((torch.rand(C, C, 3, 3, generator=g) * 2 - 1) / (3 * C ** 0.5)).cuda()
    b1 = ((torch.rand(C, generator=g) * 2 - 1) * 0.3).cuda()
    w2 = ((torch.rand(C, C, 1, 1, generator=g) * 2 - 1) / C ** 0.5).cuda()
    b2 = ((torch.rand(C, generator=g) * 2 - 1) * 0.3).cuda()
    return w1, b1, w2, b2


def run(path):
    from timbre_trap import _hip
    lib = _hip.lib()
    out = {}
    for C in (4, 8):
        for d in (1, 2, 3):
            for (B, H, T) in ((2, 13, 68), (1, 37, 36), (3, 16, 64), (1, 269 if C == 8 else 540, 1024)):
                g = torch.Generator().manual_seed(C * 100 + d)
                x = (torch.rand(B, C, H, T, generator=g) * 2 - 1).cuda()
                p = params(C, g)
                for save in (True, False):
                    y = torch.full_like(x, 7.0)
                    h = torch.full_like(x, 7.0) if save else None
                    call(lib, x, *p, y, h, d)
                    torch.cuda.synchronize()
                    out[(C, d, B, H, T, save)] = (y.cpu(), h.cpu() if save else None)
    torch.save(out, path)
    print('saved', len(out), 'cases to', path)


def cmp(a, b):
    A, Bq = torch.load(a), torch.load(b)
    bad = 0
    for k in A:
        ya, ha = A[k]
        yb, hb = Bq[k]
        if not (torch.equal(ya, yb) and (ha is None or torch.equal(ha, hb))):
            bad += 1
            print('DIFFERS', k, float((ya - yb).abs().max()))
    print('%d cases, %d differ' % (len(A), bad))
    return bad


def timeit():
    from timbre_trap import _hip
    lib = _hip.lib()
    for C, H in ((8, 269), (4, 540)):
        B, T = int(os.environ.get('KB_B', 96)), 1024
        g = torch.Generator().manual_seed(1)
        x = torch.randn(B, C, H, T, device='cuda')
        p = params(C, g)
        y = torch.empty_like(x)
        for d in (1, 2, 3):
            for _ in range(3):
                call(lib, x, *p, y, None, d)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                call(lib, x, *p, y, None, d)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print('C%d d%d fwd %.3f ms  %.2f TB/s (x + y)' % (C, d, ms, 2 * x.numel() * 4 / 1e9 / ms))


if __name__ == '__main__':
    if sys.argv[1] == 'cmp':
        sys.exit(1 if cmp(sys.argv[2], sys.argv[3]) else 0)
    elif sys.argv[1] == 'time':
        timeit()
    else:
        run(sys.argv[1])
