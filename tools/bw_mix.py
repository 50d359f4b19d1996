"""What the memory system gives a kernel that moves whole 283 MB tensors (the bf16 activations of one level at 64 clips) with nothing else to do:
1 read + 1 write, 2 + 1, 3 + 1 (the one-pass block backward: h1, dy, x in, dx out), 2 + 2, on fresh buffers each (no reuse out of the 256 MB
memory-side cache) -- the ceiling the block kernels' "achieved TB/s" figures are to be read against.  torch elementwise kernels (16 B per lane)."""
import torch
N = 64 * 65 * 1024 * 32                       # elements of one bf16 tensor at the C = 32 level
dev = 'cuda'
bufs = [torch.randn(N, device=dev, dtype=torch.bfloat16) for _ in range(12)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


state = dict(i=0)


def rot(k):
    state['i'] = (state['i'] + k) % len(bufs)
    return [bufs[(state['i'] + j) % len(bufs)] for j in range(k)]


def r1w1():
    a, o = rot(2); torch.neg(a, out=o)


def r2w1():
    a, b, o = rot(3); torch.add(a, b, out=o)


def r3w1():
    a, b, c, o = rot(4); torch.addcmul(a, b, c, out=o)


for name, fn, passes in (('1 read + 1 write', r1w1, 2), ('2 reads + 1 write', r2w1, 3), ('3 reads + 1 write', r3w1, 4)):
    ms = timeit(fn)
    print('%-18s %7.1f us  %5.2f TB/s' % (name, ms * 1e3, passes * N * 2 / (ms * 1e-3) / 1e12))
