"""Outputs of tt_wide_rb_fwd (y and h1) for a set of shapes -> a .pt file; run once per build / switch setting and compare with
   python tools/diag/wfwd_check.py cmp a.pt b.pt     (bit equality expected between the plain and the row-pipelined forward)"""
import os
import sys
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'timbre-trap_amd'))


def run(path):
    from timbre_trap import _hip
    from timbre_trap._hip import check, ptr, stream_ptr
    lib, st = _hip.lib(), stream_ptr()
    out = {}
    for C in (16, 32):
        for d in (1, 2, 3):
            for (B, H, T) in ((2, 13, 70), (1, 37, 33), (3, 16, 64), (2, 65 if C == 32 else 133, 1024)):
                g = torch.Generator().manual_seed(C * 100 + d)
                x = (torch.rand(B, H, T, C, generator=g) * 2 - 1).bfloat16().cuda()
                w1 = ((torch.rand(C, C, 3, 3, generator=g) * 2 - 1) / (3 * C ** 0.5)).cuda()
                b1 = ((torch.rand(C, generator=g) * 2 - 1) * 0.3).cuda()
                w2 = ((torch.rand(C, C, 1, 1, generator=g) * 2 - 1) / C ** 0.5).cuda()
                b2 = ((torch.rand(C, generator=g) * 2 - 1) * 0.3).cuda()
                for save in (True, False):
                    y = torch.full_like(x, 7.0)
                    h = torch.full_like(x, 7.0) if save else None
                    check(lib.tt_wide_rb_fwd(ptr(x), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(y), ptr(h), B, C, H, T, d, st), 'fwd')
                    torch.cuda.synchronize()
                    out[(C, d, B, H, T, save)] = (y.cpu(), h.cpu() if save else None)
                bad = w1.clone()
                bad[1, 2, 0, 0] = float('nan')
                y = torch.zeros_like(x)
                check(lib.tt_wide_rb_fwd(ptr(x), ptr(bad), ptr(b1), ptr(w2), ptr(b2), ptr(y), None, B, C, H, T, d, st), 'fwd')
                torch.cuda.synchronize()
                out[(C, d, B, H, T, 'nan')] = (bool(torch.isnan(y.float()).all()), None)
    torch.save(out, path)
    print('saved', len(out), 'cases to', path)


def cmp(a, b):
    A, Bq = torch.load(a), torch.load(b)
    bad = 0
    for k in A:
        ya, ha = A[k]
        yb, hb = Bq[k]
        if k[-1] == 'nan':
            ok = ya and yb
        else:
            ok = torch.equal(ya, yb) and (ha is None or torch.equal(ha, hb))
        if not ok:
            bad += 1
            print('DIFFERS', k, (float((ya.float() - yb.float()).abs().max()) if k[-1] != 'nan' else (ya, yb)))
    print('%d cases, %d differ' % (len(A), bad))
    return bad


if __name__ == '__main__':
    if sys.argv[1] == 'cmp':
        sys.exit(1 if cmp(sys.argv[2], sys.argv[3]) else 0)
    run(sys.argv[1])
