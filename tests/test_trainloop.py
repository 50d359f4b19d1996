"""
f1 row: host control flow of the train loop (timbre_trap/utils/trainloop.py) against a literal restatement of reference
experiments/train.py:336-359 (schedulers) and :534-562 (decay / best / early-stop bookkeeping) driven by the same score
sequence, plus the tensorboard tags and the checkpoint unwrap rule.
"""

import math
import os

import torch


class _Writer:
    def __init__(self):
        self.rows = []

    def add_scalar(self, tag, value, step):
        self.rows.append((tag, float(value), step))


def _reference_trace(scores, epoch_steps, interval, n_warm, n_decay, n_cool, n_early, late_start, maximize, lr=1e-3):
    """train.py:336-359 and :498-566 written out inline (the reference keeps them in one function body)."""
    from timbre_trap.utils import CosineWarmup
    lin = torch.nn.Linear(2, 2)
    opt = torch.optim.AdamW(lin.parameters(), lr=lr)
    n_cd = math.ceil(n_cool * epoch_steps / interval)
    n_dc = math.ceil(n_decay * epoch_steps / interval)
    n_es = math.ceil(n_early * epoch_steps / interval) if n_early is not None else None
    warm = CosineWarmup(opt, n_steps=n_warm * epoch_steps)
    decay = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode='max' if maximize else 'min', factor=0.5, patience=n_dc,
                                                       threshold=2E-3, cooldown=n_cd)
    best, best_ckpt, elapsed, trace, batch = None, None, 0, [], 0
    it = iter(scores)
    for i in range(1000):
        stop = False
        for _ in range(epoch_steps):
            batch += 1
            opt.step()
            if warm.is_active():
                warm.step()
            if batch % interval == 0:
                try:
                    cur = next(it)
                except StopIteration:
                    return trace
                if decay.patience and not warm.is_active() and i >= late_start:
                    decay.step(cur)
                if best is None or (maximize and cur > best) or (not maximize and cur < best):
                    best, best_ckpt, elapsed = cur, batch, 0
                else:
                    elapsed += 1
                trace.append((batch, opt.param_groups[0]['lr'], best_ckpt, elapsed))
                if n_es is not None and elapsed >= n_es:
                    stop = True
                    break
        if stop:
            trace.append('stop')
            return trace
    return trace


def _our_trace(scores, epoch_steps, interval, n_warm, n_decay, n_cool, n_early, late_start, maximize, lr=1e-3):
    from timbre_trap.utils import TrainingState, checkpoints_for, make_schedulers
    lin = torch.nn.Linear(2, 2)
    opt = torch.optim.AdamW(lin.parameters(), lr=lr)
    warm, decay = make_schedulers(opt, epoch_steps, interval, n_warm, n_decay, n_cool, maximize)
    state = TrainingState('URMP', 'mpe/f1-score', maximize,
                          checkpoints_for(n_early, epoch_steps, interval) if n_early is not None else None)
    trace, batch = [], 0
    it = iter(scores)
    for i in range(1000):
        for _ in range(epoch_steps):
            batch += 1
            opt.step()
            if warm.is_active():
                warm.step()
            if batch % interval == 0:
                try:
                    cur = next(it)
                except StopIteration:
                    return trace
                stop = state.on_checkpoint(batch, i, {'URMP': {'mpe/f1-score': cur}}, warm, decay, late_start)
                trace.append((batch, opt.param_groups[0]['lr'], state.best_model_checkpoint, state.n_checkpoints_elapsed))
                if stop:
                    trace.append('stop')
                    return trace
    return trace


def test_decay_best_and_early_stop_decisions_match_the_reference_logic():
    rising_then_flat = [0.1, 0.2, 0.3, 0.31, 0.3105, 0.3104, 0.3106, 0.3101, 0.29, 0.31, 0.3107, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3]
    configs = [
        dict(epoch_steps=10, interval=5, n_warm=1, n_decay=1, n_cool=1, n_early=3, late_start=0, maximize=True),
        dict(epoch_steps=7, interval=4, n_warm=2, n_decay=2, n_cool=0, n_early=None, late_start=3, maximize=True),
        dict(epoch_steps=10, interval=10, n_warm=0, n_decay=1, n_cool=2, n_early=4, late_start=0, maximize=False),
        dict(epoch_steps=9, interval=3, n_warm=1, n_decay=0, n_cool=0, n_early=2, late_start=0, maximize=True),     # patience 0: never stepped
    ]
    for cfg in configs:
        scores = rising_then_flat if cfg['maximize'] else [1 - s for s in rising_then_flat]
        want, got = _reference_trace(scores, **cfg), _our_trace(scores, **cfg)
        assert got == want, cfg
        assert any(r != 'stop' and r[1] < 1e-3 for r in want) or cfg['n_decay'] == 0 or cfg['n_early'] is not None


def test_step_logger_writes_the_reference_tags_with_one_transfer():
    from timbre_trap.utils import StepLogger, TRAIN_TAGS
    w = _Writer()
    losses = {'reconstruction': torch.tensor(1.5), 'transcription': torch.tensor(0.25), 'consistency/spectral': torch.tensor(2.0),
              'consistency/score': torch.tensor(3.0), 'total': torch.tensor(6.75)}
    norms = {'avg_norm/encoder': 0.1, 'max_norm/encoder': 0.2, 'avg_norm/decoder': 0.3, 'max_norm/decoder': 0.4}
    StepLogger().log(w, 17, 1e-3, losses, norms)
    assert tuple(r[0] for r in w.rows) == TRAIN_TAGS
    assert all(r[2] == 17 for r in w.rows) and w.rows[5][1] == 6.75 and w.rows[0][1] == 1e-3


def test_save_checkpoint_unwraps_data_parallel(tmp_path):
    from timbre_trap.utils import DataParallel, save_checkpoint
    lin = torch.nn.Linear(3, 2)
    path, unwrapped = save_checkpoint(DataParallel(lin), str(tmp_path), 250)
    assert os.path.basename(path) == 'model-250.pt' and unwrapped is lin
    loaded = torch.load(path, weights_only=False)
    assert isinstance(loaded, torch.nn.Linear) and torch.equal(loaded.weight, lin.weight)
