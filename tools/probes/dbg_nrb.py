import os, sys, torch
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, os.path.join(ROOT, 'timbre-trap_amd'))
from timbre_trap import _hip
from timbre_trap._hip import check, ptr, stream_ptr
lib, st = _hip.lib(), stream_ptr()
C, d = int(sys.argv[1]), int(sys.argv[2])
B, H, T = 2, {4: 540, 8: 269, 16: 133, 32: 65}[C], 1024
torch.manual_seed(0)
w1 = torch.randn(C, C, 3, 3, device='cuda') * 0.1
w2 = torch.randn(C, C, 1, 1, device='cuda') * 0.3
b1 = torch.randn(C, device='cuda') * 0.1
b2 = torch.randn(C, device='cuda') * 0.1
xb = torch.randn(B, H, T, C, device='cuda').bfloat16()
gb = torch.randn(B, H, T, C, device='cuda').bfloat16()
yb, hb, dxb = (torch.empty_like(xb) for _ in range(3))
check(lib.tt_wide_rb_fwd(ptr(xb), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(yb), ptr(hb), B, C, H, T, d, st), 'fwd')
ws = torch.zeros(lib.tt_wide_scratch_bytes(B, C, H, T), dtype=torch.uint8, device='cuda')
res = []
for it in range(12):
    g = [torch.zeros(s, dtype=torch.float32, device='cuda') for s in ((C, C, 3, 3), (C,), (C, C, 1, 1), (C,))]
    check(lib.tt_wide_rb_bwd(ptr(xb), ptr(hb), ptr(gb), ptr(w1), ptr(w2), ptr(b2), ptr(dxb), ptr(g[0]), ptr(g[1]), ptr(g[2]), ptr(g[3]), ptr(ws), B, C, H, T, d, st), 'bwd')
    torch.cuda.synchronize()
    res.append([t.clone() for t in g] + [dxb.clone(), ws.clone()])
ref = res[0]
for it in range(1, 12):
    r = res[it]
    msg = []
    for k, name in enumerate(('dw1', 'db1', 'dw2', 'db2', 'dx')):
        dif = (r[k].float() - ref[k].float()).abs()
        if float(dif.max()) > 0:
            idx = torch.nonzero(dif.reshape(-1) > 0).reshape(-1)[:6].tolist()
            msg.append('%s: %d differ, max %.3e at %s' % (name, int((dif > 0).sum()), float(dif.max()), idx))
    # per-workgroup partials (part_a region starts at the front of the scratch after dA1?) -- report the float view differences
    wsd = (r[5].view(torch.float32) - ref[5].view(torch.float32)).abs()
    nz = torch.nonzero(wsd > 0).reshape(-1)
    msg.append('ws floats differing: %d first %s' % (nz.numel(), nz[:8].tolist()))
    if len(msg) > 1: print(it, '; '.join(msg))
print('C %d d %d: distinct dw2[0,0] values over 12 runs: %d' % (C, d, len(set(float(r[2].reshape(-1)[0]) for r in res))))
# per-workgroup partials of dW2[0][0]: part_a follows the dA1 region of the scratch
ADUMP = C * C + 2 * C
base = B * H * T * C * 2 // 4
pa = [r[5].view(torch.float32)[base:base + 2048 * ADUMP].view(2048, ADUMP)[:, 0].clone() for r in res]
ref_pa = torch.stack(pa).median(0).values
for it in range(4):
    d = pa[it] - ref_pa
    nz = torch.nonzero(d != 0).reshape(-1)
    print('run %d: %d workgroups off: %s' % (it, nz.numel(), [(int(j), round(float(ref_pa[j]), 3), round(float(d[j]), 3)) for j in nz[:10]]))
