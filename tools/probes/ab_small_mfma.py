import os, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'timbre-trap_amd')
from timbre_trap.framework import ops
torch.manual_seed(0)
for C in (4, 8):
    x = torch.randn(3, C, 45, 200, device='cuda')
    w1, b1 = torch.randn(C, C, 3, 3, device='cuda') * 0.2, torch.randn(C, device='cuda') * 0.2
    w2, b2 = torch.randn(C, C, 1, 1, device='cuda') * 0.3, torch.randn(C, device='cuda') * 0.2
    for d in (1, 2, 3):
        outs = {}
        for mode in ('mfma', 'valu'):
            os.environ.pop('TTRAP_SMALL_VALU_FMA', None)
            if mode == 'valu':
                os.environ['TTRAP_SMALL_VALU_FMA'] = '1'
            os.environ['TTRAP_SMALL_UNFUSED_BWD'] = '1'          # dgrad through k_small_lds MODE 1 as well
            xr = x.clone().requires_grad_(True)
            y = ops.ResBlockFn.apply(xr, w1, b1, w2, b2, d)
            gx, = torch.autograd.grad(y, xr, torch.ones_like(y))
            outs[mode] = (y.detach().clone(), gx.clone())
        print('C=%d d=%d  forward bitwise equal: %s   dx bitwise equal: %s' % (C, d, torch.equal(outs['mfma'][0], outs['valu'][0]), torch.equal(outs['mfma'][1], outs['valu'][1])))
