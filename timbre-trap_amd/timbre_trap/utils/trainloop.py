"""
Host side of the training loop around the hot path: the control flow of reference
``experiments/train.py:333-566`` that is not device work -- learning-rate schedule wiring (:336-359), per-step scalar logging
with the reference's tensorboard tags (:401-490), checkpointing (:502-511), and the best-checkpoint / plateau-decay /
early-stopping bookkeeping (:534-566).  The reference keeps all of this inline in a sacred ``@ex.automain`` function; here it
is a small set of objects so that a driver script (or the reference's own script, edited to call them) gets identical
decisions: same checkpoint counts, same decay steps, same best checkpoint, same stop iteration.

One difference by design: ``StepLogger`` collects the step's device scalars and reads them back with ONE host sync instead of
one ``.item()`` per tag (the reference issues ~19 per step).
"""

import math
import os

import torch

from .experiments import CosineWarmup

__all__ = ['checkpoints_for', 'make_schedulers', 'StepLogger', 'TrainingState', 'save_checkpoint', 'print_and_log',
           'log_gradient_norms', 'TRAIN_TAGS']

# tensorboard tags of reference experiments/train.py:401-490, in the order they are written
TRAIN_TAGS = ('train/loss/learning_rate', 'train/loss/reconstruction', 'train/loss/transcription',
              'train/loss/consistency/spectral', 'train/loss/consistency/score', 'train/loss/total',
              'train/avg_norm/encoder', 'train/max_norm/encoder', 'train/avg_norm/decoder', 'train/max_norm/decoder')


def checkpoints_for(n_epochs, epoch_steps, checkpoint_interval):
    """Validation checkpoints spanned by ``n_epochs`` epochs (train.py:340-345): ceil(n_epochs * epoch_steps / interval)."""
    return math.ceil(n_epochs * epoch_steps / checkpoint_interval)


def make_schedulers(optimizer, epoch_steps, checkpoint_interval, n_epochs_warmup, n_epochs_decay, n_epochs_cooldown,
                    maximize=True):
    """
    (warmup, decay) exactly as train.py:350-359 builds them: a reverse-cosine warm-up over ``n_epochs_warmup * epoch_steps``
    optimiser steps, and ReduceLROnPlateau(factor 0.5, threshold 2e-3) whose patience and cooldown are counted in validation
    checkpoints.
    """
    warmup = CosineWarmup(optimizer, n_steps=n_epochs_warmup * epoch_steps)
    decay = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode='max' if maximize else 'min', factor=0.5,
                                                       patience=checkpoints_for(n_epochs_decay, epoch_steps, checkpoint_interval),
                                                       threshold=2E-3,
                                                       cooldown=checkpoints_for(n_epochs_cooldown, epoch_steps, checkpoint_interval))
    return warmup, decay


class StepLogger:
    """
    ``log(writer, batch_count, lr, losses, norms)`` writes the reference's per-step tags.  ``losses`` maps the tag suffix after
    ``train/loss/`` to a device scalar (``reconstruction``, ``transcription``, ``consistency/spectral``, ``consistency/score``,
    any model-specific extras of train.py:459-462, ``total``); all of them come back in one transfer.
    """

    def log(self, writer, batch_count, learning_rate, losses, norms=None):
        writer.add_scalar('train/loss/learning_rate', learning_rate, batch_count)
        keys = list(losses.keys())
        if keys:
            stacked = torch.stack([losses[k].detach().reshape(()).float() for k in keys])
            for k, v in zip(keys, stacked.cpu().tolist()):                    # the only host sync of the step's logging
                writer.add_scalar('train/loss/' + k, v, batch_count)
        for k, v in (norms or {}).items():                                    # e.g. 'avg_norm/encoder' (already floats)
            writer.add_scalar('train/' + k, float(v), batch_count)


def print_and_log(text, path=None):
    """Print, and append to a text file when a path is given (reference utils/experiments.py:44-64)."""
    print(text)
    if path is not None:
        with open(path, 'a') as f:
            print(text, file=f)


def log_gradient_norms(module, writer, i=0, prefix='gradients/norm'):
    """One scalar per parameter gradient (reference utils/experiments.py:258-277), from one kernel launch and one host sync."""
    from .experiments import gradient_statistics
    norms, _ = gradient_statistics(module)
    names = [n for n, p in module.named_parameters() if p.grad is not None]
    for layer, v in zip(names, norms):
        writer.add_scalar('%s/%s' % (prefix, layer), float(v), i)


def save_checkpoint(model, log_dir, batch_count):
    """
    ``torch.save`` of the whole (unwrapped) module to ``<log_dir>/model-<batch_count>.pt`` (train.py:502-511).
    Returns (path, unwrapped model) -- the caller re-wraps for multi-GPU use like train.py:525-527.
    """
    path = os.path.join(log_dir, 'model-%d.pt' % batch_count)
    if isinstance(model, torch.nn.DataParallel):
        model = model.module
    torch.save(model, path)
    return path, model


class TrainingState:
    """
    Best-checkpoint / plateau / early-stop bookkeeping of train.py:364-566.

        state = TrainingState(criteria_set, criteria_metric, maximize, n_checkpoints_early_stop)
        stop = state.on_checkpoint(batch_count, epoch_index, validation_results, warmup, decay, n_epochs_late_start)

    ``validation_results`` is {dataset name: {metric: value}} as returned by evaluate().  The decay scheduler is stepped only
    when it has a patience, the warm-up has finished and the late-start epoch has been reached (train.py:534-536); ``best``
    is replaced on STRICT improvement (:544-552); ``stop`` turns True once ``n_checkpoints_early_stop`` checkpoints in a row
    have not improved (:558-562).
    """

    def __init__(self, criteria_set, criteria_metric, maximize=True, n_checkpoints_early_stop=None):
        self.criteria_set, self.criteria_metric, self.maximize = criteria_set, criteria_metric, maximize
        self.n_checkpoints_early_stop = n_checkpoints_early_stop
        self.best_model_checkpoint = None
        self.best_results = None
        self.n_checkpoints_elapsed = 0
        self.early_stop_criteria = False

    def on_checkpoint(self, batch_count, epoch_index, validation_results, warmup_scheduler, decay_scheduler, n_epochs_late_start=0):
        current = validation_results[self.criteria_set][self.criteria_metric]
        if decay_scheduler.patience and not warmup_scheduler.is_active() and epoch_index >= n_epochs_late_start:
            decay_scheduler.step(current)
        best = None if self.best_results is None else self.best_results[self.criteria_set][self.criteria_metric]
        if best is None or (self.maximize and current > best) or (not self.maximize and current < best):
            self.best_model_checkpoint = batch_count
            self.best_results = validation_results
            self.n_checkpoints_elapsed = 0
        else:
            self.n_checkpoints_elapsed += 1
        if self.n_checkpoints_early_stop is not None and self.n_checkpoints_elapsed >= self.n_checkpoints_early_stop:
            self.early_stop_criteria = True
        return self.early_stop_criteria
