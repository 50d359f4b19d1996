"""Time the six latent-layer GEMM shapes of one train step through the C ABI (B = 64, T = 1024)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'timbre-trap_amd'))
import torch
from timbre_trap import _hip
from timbre_trap._hip import ptr, stream_ptr
lib = _hip.lib()
B, T, D, K, Kd = 64, 1024, 128, 1984, 129
dev = 'cuda'
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
w = torch.randn(D, K, device=dev); x = torch.randn(B, K, T, device=dev); y = torch.empty(B, D, T, device=dev); dy = torch.randn(B, D, T, device=dev)
dx = torch.empty_like(x); dw = torch.zeros_like(w)
wd = torch.randn(Kd, K, device=dev); z = torch.randn(B, Kd, T, device=dev); yd = torch.empty(B, K, T, device=dev); g = torch.randn(B, K, T, device=dev)
dz = torch.empty_like(z); dwd = torch.zeros_like(wd)
N = None
cases = [
 ('enc fwd   M=128  N=1024 K=1984', lambda: lib.tt_gemm(ptr(w), ptr(x), ptr(y), N, D, T, K, 0, 0, K, T, T, B, 0, K * T, D * T, 0, 1.0, 0.0, 0, 1, 0, stream_ptr())),
 ('enc dgrad M=1984 N=1024 K=128 ', lambda: lib.tt_gemm(ptr(w), ptr(dy), ptr(dx), N, K, T, D, 1, 0, K, T, T, B, 0, D * T, K * T, 0, 1.0, 0.0, 0, 1, 0, stream_ptr())),
 ('enc wgrad M=128  N=1984 K=1024', lambda: lib.tt_gemm(ptr(dy), ptr(x), ptr(dw), N, D, K, T, 0, 1, T, T, K, B, D * T, K * T, 0, 1, 1.0, 1.0, 0, 1, 0, stream_ptr())),
 ('dec fwd   M=1984 N=1024 K=129 ', lambda: lib.tt_gemm(ptr(wd), ptr(z), ptr(yd), N, K, T, Kd, 1, 0, K, T, T, B, 0, Kd * T, K * T, 0, 1.0, 0.0, 0, 31, 0, stream_ptr())),
 ('dec dgrad M=129  N=1024 K=1984', lambda: lib.tt_gemm(ptr(wd), ptr(g), ptr(dz), N, Kd, T, K, 0, 0, K, T, T, B, 0, K * T, Kd * T, 0, 1.0, 0.0, 0, 1, 0, stream_ptr())),
 ('dec wgrad M=129  N=1984 K=1024', lambda: lib.tt_gemm(ptr(z), ptr(g), ptr(dwd), N, Kd, K, T, 0, 1, T, T, K, B, Kd * T, K * T, 0, 1, 1.0, 1.0, 0, 1, 0, stream_ptr())),
]
for name, fn in cases:
    ms = t(fn)
    print('%s  %.3f ms  %5.1f TFLOP/s' % (name, ms, 2.0 * 128 * 1984 * 1024 * 64 / ms / 1e9))
