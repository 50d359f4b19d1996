"""
Host side of the train loop next to the hot path: the pieces of reference
``timbre_trap/utils/experiments.py`` that ``experiments/train.py`` calls every step.

  * ``CosineWarmup``            same closed form and stepping protocol as the reference scheduler (:81-141)
  * gradient statistics         ``sum_gradient_norms / average_gradient_norms / get_max_gradient / get_max_gradient_norm``
                                (:144-256).  The reference does one ``.item()`` host sync per layer and per statistic
                                (~480 per step); here every per-parameter norm and abs-max comes from ONE kernel over
                                the flat gradient buffer (tt_segment_stats) and ONE device->host copy per call.
  * ``DataParallel``            re-exported from .distributed (one process per GPU).
"""

import ctypes
import random
from math import cos, pi

import numpy as np
import torch

from .. import _hip
from .distributed import DataParallel

__all__ = ['seed_everything', 'DataParallel', 'CosineWarmup', 'sum_gradient_norms', 'average_gradient_norms',
           'get_max_gradient', 'get_max_gradient_norm', 'gradient_statistics']


def seed_everything(seed):
    """Seed Python, NumPy and torch (reference experiments.py:29-49)."""
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


class CosineWarmup(torch.optim.lr_scheduler.LRScheduler):
    """Reverse-cosine learning-rate warm-up over ``n_steps`` optimiser steps."""

    def __init__(self, optimizer, n_steps):
        self.n_steps = max(0, n_steps)
        super().__init__(optimizer)

    def is_active(self):
        return self.last_epoch < self.n_steps

    def reset(self):
        self.last_epoch = -1
        self.step()

    def scale_at(self, step_index):
        curr_step = 1 + min(step_index, self.n_steps)
        return 1 - 0.5 * (1 + cos(curr_step * pi / (self.n_steps + 1)))

    def get_lr(self):
        return self._get_closed_form_lr()

    def _get_closed_form_lr(self):
        scaling = self.scale_at(self.last_epoch)
        return [scaling * base_lr for base_lr in self.base_lrs]


def gradient_statistics(module):
    """
    (norms, absmax): two float64 NumPy arrays with the L2 norm and the max magnitude of every parameter gradient of
    ``module`` (in ``named_parameters`` order, parameters without a gradient skipped) -- one launch, one host sync.
    """
    grads = [p.grad for _, p in module.named_parameters() if p.grad is not None]
    if not grads:
        return np.zeros(0), np.zeros(0)
    _hip.require_cuda(*grads)
    storage = grads[0].untyped_storage().data_ptr()
    flat_views = all(g.dtype == torch.float32 and g.is_contiguous() and g.untyped_storage().data_ptr() == storage for g in grads)
    if flat_views:
        # FusedAdamW keeps every .grad as a view of one flat buffer: address it in place
        base = storage
        begins = [(g.data_ptr() - storage) // 4 for g in grads]
        keep = None
    else:
        keep = torch.cat([g.detach().reshape(-1).to(torch.float32) for g in grads])
        base = keep.data_ptr()
        begins = list(np.cumsum([0] + [g.numel() for g in grads[:-1]]))
    spans = torch.tensor([[int(b), int(b) + g.numel()] for b, g in zip(begins, grads)], dtype=torch.int64).to(grads[0].device)
    out = torch.empty(len(grads), 2, dtype=torch.float32, device=grads[0].device)
    _hip.check(_hip.lib().tt_segment_stats(ctypes.c_void_p(base), _hip.ptr(spans), len(grads), _hip.ptr(out),
                                           _hip.stream_ptr()), 'tt_segment_stats')
    res = out.cpu().double().numpy()
    del keep
    return res[:, 0], res[:, 1]


def sum_gradient_norms(module):
    return float(gradient_statistics(module)[0].sum())


def average_gradient_norms(module):
    norms = gradient_statistics(module)[0]
    return float(norms.sum() / len(norms))


def get_max_gradient(module):
    amax = gradient_statistics(module)[1]
    return float(amax.max()) if len(amax) else 0.


def get_max_gradient_norm(module):
    norms = gradient_statistics(module)[0]
    return float(norms.max()) if len(norms) else 0.
