"""
Fused gradient clipping + AdamW over one flat fp32 buffer (csrc/losses.hip: tt_l2norm, tt_adamw_step).

Replaces ``torch.nn.utils.clip_grad_norm_(model.parameters(), 10)`` + ``torch.optim.AdamW.step()``
of reference experiments/train.py:334 and :493-496 (614 k parameters in ~120 tensors -> two kernel
launches instead of several hundred).  Parameters stay ordinary ``nn.Parameter`` objects: they are
re-pointed at slices of one flat buffer, and ``.grad`` at slices of one flat gradient buffer, so the
reference's grad-norm logging (utils/experiments.py:172-256), ``clip_grad_norm_`` and any stock
optimiser keep working on the same objects, and the data-parallel all-reduce is ONE collective.
"""

import torch

from .. import _hip


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_norm=None):
        params = [p for p in params if p.requires_grad]
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise ValueError('FusedAdamW keeps one flat buffer: a single param group only')
        self.max_norm = max_norm
        self._flatten(params)
        self._step = 0

    def _flatten(self, params):
        dev = params[0].device
        _hip.require_cuda(*params)
        n = sum(p.numel() for p in params)
        self.flat_param = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self._partials = torch.empty(1024, dtype=torch.float64, device=dev)
        o = 0
        with torch.no_grad():
            for p in params:
                k = p.numel()
                self.flat_param[o:o + k].copy_(p.detach().reshape(-1))
                p.data = self.flat_param[o:o + k].view_as(p)
                p.grad = self.flat_grad[o:o + k].view_as(p)
                p._ttrap_accumulate = True        # backward kernels add into this view directly (framework/ops.py:_grad_target)
                o += k
        self.n = n

    def zero_grad(self, set_to_none=False):
        # gradients must stay views of the flat buffer: zero in place, never drop them
        self.flat_grad.zero_()

    def grad_norm(self):
        _hip.check(_hip.lib().tt_l2norm(_hip.ptr(self.flat_grad), _hip.ptr(self.norm), _hip.ptr(self._partials), self.n,
                                        _hip.stream_ptr()), 'tt_l2norm')
        return self.norm

    @torch.no_grad()
    def step(self, closure=None):
        """Clip (if max_norm) and apply AdamW.  Returns the pre-clip gradient norm (device tensor, no sync)."""
        if closure is not None:
            raise NotImplementedError
        g = self.param_groups[0]
        self._step += 1
        norm = self.grad_norm() if self.max_norm else None
        _hip.check(_hip.lib().tt_adamw_step(_hip.ptr(self.flat_param), _hip.ptr(self.flat_grad), _hip.ptr(self.exp_avg),
                                            _hip.ptr(self.exp_avg_sq), _hip.ptr(norm), self.n, float(g['lr']),
                                            float(g['betas'][0]), float(g['betas'][1]), float(g['eps']),
                                            float(g['weight_decay']), self._step,
                                            float(self.max_norm) if self.max_norm else 0.0, 1, _hip.stream_ptr()),
                   'tt_adamw_step')
        return self.norm if self.max_norm else None
