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


def _mark_touched(p):
    p._ttrap_touched = True


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_norm=None):
        params = [p for p in params if p.requires_grad]
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise ValueError('FusedAdamW keeps one flat buffer: a single param group only')
        self.max_norm = max_norm
        # slots whose .grad was None when _reattach() last met them; consumed by step(), cleared by zero_grad(), and overruled by a
        # gradient that arrives afterwards (``_ttrap_touched``: set where a backward kernel is handed the view, ops._grad_target, and by
        # autograd's own accumulation, the hook below)
        self._nograd = set()
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
        # steps skipped because the gradient norm was not finite (device counter, no host sync): the overflow guard that goes with the
        # static loss scale of the fp16 path (ops.FP16_LOSS_SCALE) -- what torch.amp.GradScaler does on inf / NaN gradients, and what
        # the reference, which has autocast without a scaler (train.py:415), lacks.  Needs max_norm (the norm is the detector).
        self.skipped = torch.zeros(1, dtype=torch.int32, device=dev)
        self._partials = torch.empty(1024, dtype=torch.float64, device=dev)
        o = 0
        self._slots = []
        with torch.no_grad():
            for p in params:
                k = p.numel()
                self.flat_param[o:o + k].copy_(p.detach().reshape(-1))
                p.data = self.flat_param[o:o + k].view_as(p)
                p.grad = self.flat_grad[o:o + k].view_as(p)
                p._ttrap_accumulate = True        # backward kernels add into this view directly (framework/ops.py:_grad_target)
                p._ttrap_touched = False
                p.register_post_accumulate_grad_hook(_mark_touched)       # gradients that autograd itself adds into the view
                self._slots.append((p, o, k))
                o += k
        self.n = n

    def _reattach(self):
        """
        The fused step reads ONE flat gradient buffer and writes ONE flat parameter buffer, so every ``p.grad`` / ``p.data``
        must still be the view it was given.  ``model.zero_grad()`` (grads set to None), ``p.grad = None``, or anything else
        that makes autograd install a fresh ``.grad`` tensor detaches a view: the fresh gradient is then copied into its slot
        and the view restored (a parameter without a gradient contributes zeros).  A re-allocated ``p.data`` on the same
        device is adopted the same way; one on another device is an error (build a new optimizer after moving the model).

        A slot found with ``p.grad is None`` is REMEMBERED (``self._nograd``) until ``step()`` consumes it: re-attaching replaces the
        None by a zero view, and ``sync_views()`` / ``grad_norm()`` / ``GradientSync.start(opt)`` all re-attach before ``step()``
        runs -- without the record such a parameter would take weight decay and residual momentum, which ``torch.optim.AdamW`` skips.
        The record only says "no gradient SO FAR": a re-attachment between ``model.zero_grad()`` (gradients set to None) and the
        backward pass -- ``grad_norm()`` logged at the end of the previous iteration, ``sync_views()`` -- would otherwise mark EVERY
        slot, and the backward that follows accumulates straight into the restored views: a gradient that arrives after the record
        was made (``p._ttrap_touched``) takes the slot out of it again (round-5 advisor finding), and so does ``zero_grad()``.
        """
        gbase, pbase = self.flat_grad.data_ptr(), self.flat_param.data_ptr()
        for p, o, k in self._slots:
            g = p.grad
            if g is None or g.data_ptr() != gbase + 4 * o or g.dtype != torch.float32:
                slot = self.flat_grad[o:o + k]
                if g is None:
                    slot.zero_()
                    self._nograd.add(o)
                    p._ttrap_touched = False          # whatever arrives from here on is a gradient of this step
                else:
                    if g.device != slot.device:
                        raise RuntimeError('FusedAdamW: a gradient moved to %s (optimizer state is on %s)' % (g.device, slot.device))
                    slot.copy_(g.detach().reshape(-1))
                    self._nograd.discard(o)           # a fresh gradient installed by autograd since the record was made
                p.grad = slot.view_as(p)
            if p.data_ptr() != pbase + 4 * o:
                if p.device != self.flat_param.device or p.dtype != torch.float32:
                    raise RuntimeError('FusedAdamW: parameter storage was replaced (device %s, dtype %s); create a new '
                                       'optimizer after moving or casting the model' % (p.device, p.dtype))
                self.flat_param[o:o + k].copy_(p.detach().reshape(-1))
                p.data = self.flat_param[o:o + k].view_as(p)
            p._ttrap_accumulate = True

    def sync_views(self):
        """
        Public form of the re-attachment: call it (or anything below that does: ``grad_norm``, ``step``, ``GradientSync.start(opt)``)
        BEFORE reading ``flat_grad`` whenever ``model.zero_grad()`` / ``p.grad = None`` may have detached a gradient view since
        the last step -- in particular before the data-parallel all-reduce, which must see this step's gradient in the flat
        buffer, not the stale slot.  Returns the flat gradient buffer.
        """
        self._reattach()
        return self.flat_grad

    def state_dict(self):
        """Moments and step count included (they live in flat buffers, outside ``Optimizer.state``)."""
        sd = super().state_dict()
        sd['ttrap_flat'] = dict(step=self._step, exp_avg=self.exp_avg.detach().clone(), exp_avg_sq=self.exp_avg_sq.detach().clone(),
                                max_norm=self.max_norm, n=self.n, skipped=int(self.skipped.item()))
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        flat = state_dict.pop('ttrap_flat', None)
        super().load_state_dict(state_dict)
        if flat is None:
            raise ValueError('not a FusedAdamW state_dict (no flat moments): resuming would restart the moment estimates')
        if int(flat['n']) != self.n:
            raise ValueError('FusedAdamW state has %d elements, this optimizer %d' % (int(flat['n']), self.n))
        self._step = int(flat['step'])
        self.skipped.fill_(int(flat.get('skipped', 0)))
        self.exp_avg.copy_(flat['exp_avg'].to(self.exp_avg.device))
        self.exp_avg_sq.copy_(flat['exp_avg_sq'].to(self.exp_avg_sq.device))

    def zero_grad(self, set_to_none=False):
        # gradients must stay views of the flat buffer: zero in place, never drop them (and restore dropped ones)
        self.flat_grad.zero_()
        self._nograd.clear()               # every slot is a (zero) view again: "had no gradient" starts over with this step
        for p, o, k in self._slots:
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                p.grad = self.flat_grad[o:o + k].view_as(p)

    def grad_norm(self):
        self._reattach()
        _hip.check(_hip.lib().tt_l2norm(_hip.ptr(self.flat_grad), _hip.ptr(self.norm), _hip.ptr(self._partials), self.n,
                                        _hip.stream_ptr()), 'tt_l2norm')
        return self.norm

    @torch.no_grad()
    def step(self, closure=None):
        """Clip (if max_norm) and apply AdamW.  Returns the pre-clip gradient norm (device tensor, no sync)."""
        if closure is not None:
            raise NotImplementedError
        g = self.param_groups[0]
        # torch.optim.AdamW skips a parameter without a gradient (frozen after the optimizer was built, or ``p.grad = None``): neither
        # its residual momentum nor the weight decay may move it.  The fused kernel walks the whole flat buffer, so the slots of such
        # parameters (zero gradient: nothing in the clipping norm either) are put back, moments included, after the update.
        self._reattach()                   # records the slots whose gradient is None (now, or at an earlier re-attachment of this step)
        frozen = [(o, k) for p, o, k in self._slots if (o in self._nograd and not getattr(p, '_ttrap_touched', False)) or not p.requires_grad]
        self._nograd.clear()
        keep = [(o, k, self.flat_param[o:o + k].clone(), self.exp_avg[o:o + k].clone(), self.exp_avg_sq[o:o + k].clone()) for o, k in frozen]
        for o, k in frozen:
            self.flat_grad[o:o + k].zero_()
        self._step += 1
        with _hip.timed('clip_adamw'):
            norm = self._clip_and_update(g)
        for o, k, p0, m0, v0 in keep:
            self.flat_param[o:o + k].copy_(p0)
            self.exp_avg[o:o + k].copy_(m0)
            self.exp_avg_sq[o:o + k].copy_(v0)
        return norm

    def _clip_and_update(self, g):
        norm = self.grad_norm() if self.max_norm else None
        _hip.check(_hip.lib().tt_adamw_step(_hip.ptr(self.flat_param), _hip.ptr(self.flat_grad), _hip.ptr(self.exp_avg),
                                            _hip.ptr(self.exp_avg_sq), _hip.ptr(norm), self.n, float(g['lr']),
                                            float(g['betas'][0]), float(g['betas'][1]), float(g['eps']),
                                            float(g['weight_decay']), self._step,
                                            float(self.max_norm) if self.max_norm else 0.0, 1,
                                            _hip.ptr(self.skipped) if self.max_norm else None, _hip.stream_ptr()),
                   'tt_adamw_step')
        return self.norm if self.max_norm else None
