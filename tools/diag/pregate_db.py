"""debug: bias gradient of tt_tconv16_bwd_pregated per channel at C = 32."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'timbre-trap_amd')):
    sys.path.insert(0, p)
import torch
from timbre_trap import _hip
from timbre_trap._hip import check, ptr, stream_ptr
from timbre_trap.framework import ops
lib, st = _hip.lib(), stream_ptr()
C = int(os.environ.get('DBG_C', 32))
for (B, H, T, out_pad) in [(1, 4, 64, 0), (1, 4, 64, 1), (1, 5, 64, 0), (2, 5, 80, 0), (1, 8, 64, 0)]:
    Ho = 2 * H + 2 + out_pad
    torch.manual_seed(0)
    x = torch.randn(B, 2 * C, H, T)
    g = torch.randn(B, C, Ho, T)
    w = torch.randn(2 * C, C, 4, 1).cuda() * 0.1
    xb, gb = ops._pack(x.cuda(), torch.bfloat16), ops._pack(g.cuda(), torch.bfloat16)
    ws = torch.empty(lib.tt_stride16_scratch_bytes(C), dtype=torch.uint8, device='cuda')
    for with_dx in (True, False):
        dx = ops.new_cl16(B, 2 * C, H, T, 'cuda', torch.bfloat16)
        dw, db = torch.zeros(2 * C, C, 4, 1, device='cuda'), torch.zeros(C, device='cuda')
        check(lib.tt_tconv16_bwd_pregated(ptr(xb), ptr(gb), ptr(w), ptr(dx) if with_dx else None, ptr(dw), ptr(db), ptr(ws), B, C, H, T, out_pad, st), 'x')
        want = gb.float().sum((0, 2, 3))
        # per-row sums to see which rows are missing / doubled
        rows = gb.float().sum((0, 3))           # (C, Ho)
        err = (db - want).cpu()
        print('shape', (B, H, T, out_pad), 'dx' if with_dx else 'nodx', 'max err per 16-tile', [float(err[i:i + 16].abs().max()) for i in range(0, C, 16)])
        if float(err.abs().max()) > 1e-2:
            ch = int(err.abs().argmax())
            # solve which rows explain the error: err = sum_r coef_r * rows[ch, r]
            A = rows[:, :].cpu().T                                  # (Ho, C) -> unknown coef per row, same for all channels of the tile
            t0 = (ch // 16) * 16
            coef = torch.linalg.lstsq(A[:, t0:t0 + 16].T.double(), err[t0:t0 + 16].double().unsqueeze(1)).solution.squeeze()
            print('   row coefficients (0 = counted once):', [round(float(v), 2) for v in coef])
