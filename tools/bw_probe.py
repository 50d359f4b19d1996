"""Achievable HBM stream rates on this GPU for the resblock tensor size (torch elementwise kernels, 16 B/lane)."""
import torch
x = torch.randn(64, 32, 65, 1024, device='cuda')
y = torch.empty_like(x)
z = torch.empty_like(x)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
nb = x.numel() * 4
ms = t(lambda: y.copy_(x)); print('copy       %.3f ms  %.2f TB/s' % (ms, 2 * nb / ms / 1e9))
ms = t(lambda: torch.add(x, 1.0, out=y)); print('add        %.3f ms  %.2f TB/s' % (ms, 2 * nb / ms / 1e9))
ms = t(lambda: torch.add(x, z, out=y)); print('add2       %.3f ms  %.2f TB/s (2R 1W)' % (ms, 3 * nb / ms / 1e9))
ms = t(lambda: y.fill_(1.0)); print('fill       %.3f ms  %.2f TB/s (write only)' % (ms, nb / ms / 1e9))
ms = t(lambda: x.sum()); print('sum        %.3f ms  %.2f TB/s (read only)' % (ms, nb / ms / 1e9))
def two():
    y.copy_(x); z.fill_(0.0)
ms = t(two); print('copy+fill  %.3f ms  %.2f TB/s (1R 2W, two kernels)' % (ms, 3 * nb / ms / 1e9))
