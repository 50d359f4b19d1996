"""A/B of the narrow-level residual-block backward at the bench shapes: python tools/kb_small_bwd.py  (env KB_C, KB_B)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'timbre-trap_amd'))
import torch  # noqa: E402
from timbre_trap import _hip  # noqa: E402
from timbre_trap._hip import ptr  # noqa: E402

LEVELS = {4: 540, 8: 269}


def main():
    lib = _hip.lib()
    B, T = int(os.environ.get('KB_B', 64)), 1024
    st = _hip.stream_ptr()
    for C, H in LEVELS.items():
        if os.environ.get('KB_C') and int(os.environ['KB_C']) != C:
            continue
        x, h1, dy = (torch.randn(B, C, H, T, device='cuda') for _ in range(3))
        dx = torch.empty_like(x)
        w1, b1 = torch.randn(C, C, 3, 3, device='cuda') * 0.1, torch.randn(C, device='cuda') * 0.1
        w2, b2 = torch.randn(C, C, 1, 1, device='cuda') * 0.1, torch.randn(C, device='cuda') * 0.1
        dw1, db1, dw2, db2 = torch.zeros_like(w1), torch.zeros_like(b1), torch.zeros_like(w2), torch.zeros_like(b2)
        ws = torch.empty(x.numel() + lib.tt_wgrad_scratch_floats(), device='cuda')
        for d in (1, 3):
            def run():
                _hip.check(lib.tt_resblock_bwd(ptr(x), ptr(h1), ptr(dy), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(dx), ptr(dw1), ptr(db1),
                                               ptr(dw2), ptr(db2), ptr(ws), B, C, H, T, d, 0, st))
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                run()
            e.record()
            torch.cuda.synchronize()
            ms = s.elapsed_time(e) / 10
            print('small_bwd C=%d d=%d  %.3f ms   (4 tensors = %.0f GB/s)  ablate=%s unfused=%s' % (
                C, d, ms, 4 * x.numel() * 4 / ms / 1e6, os.environ.get('TTRAP_FUSED_ABLATE', '0'), os.environ.get('TTRAP_SMALL_UNFUSED_BWD', '0')))


if __name__ == '__main__':
    main()
