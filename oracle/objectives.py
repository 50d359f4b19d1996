"""
ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of reference ``timbre_trap/framework/objectives.py`` written with explicit
reductions (no ``mse_loss``), plus the train-step arithmetic of
``experiments/train.py:404-496``.  Pinned by tests/golden/objectives.npz.
"""

import torch


def compute_reconstruction_loss(reconstructed, target):
    """objectives.py:11-33 : squared error summed over (channel, F), averaged over (B, T)."""
    err = (reconstructed - target) ** 2
    return err.sum(-3).sum(-2).mean()


def compute_transcription_loss(estimate, target, weight_positive_class=False):
    """
    objectives.py:36-74.  With weighting: per frame scale = sum_F(1-tgt) / (sum_F tgt + eps32),
    applied only where tgt == 1 exactly, every other element (and a zero scale) weighs 1.
    """
    err = (estimate - target) ** 2
    if weight_positive_class:
        pos = target.sum(dim=-2, keepdim=True)
        neg = (1 - target).sum(dim=-2, keepdim=True)
        scale = neg / (pos + torch.finfo(torch.float32).eps)
        scaling = scale * (target == 1)
        scaling = torch.where(scaling == 0, torch.ones_like(scaling), scaling)
        err = err * scaling
    return err.sum(-2).mean()


def compute_consistency_loss(spectral_coefficients, transcription_coefficients, target):
    """objectives.py:77-104 : two reconstruction losses against the (non-detached) target."""
    return (compute_reconstruction_loss(spectral_coefficients, target),
            compute_reconstruction_loss(transcription_coefficients, target))


def total_loss(outputs, coefficients, ground_truth, multipliers=None, n_mpe=None):
    """
    train.py:418-464 after the late start: total = m_rec*rec + m_trn*trn + m_con*(con_sp + con_sc).
    ``outputs`` = (reconstruction, latents, transcription, transcription_rec, transcription_scr).
    """
    from .autoencoder import to_activations
    m = dict(reconstruction=1, transcription=1, consistency=1)
    if multipliers:
        m.update(multipliers)
    rec, _, trn_coeffs, trn_rec, trn_scr = outputs
    n = ground_truth.size(0) if n_mpe is None else n_mpe
    activations = to_activations(trn_coeffs)
    parts = dict(reconstruction=compute_reconstruction_loss(rec, coefficients),
                 transcription=compute_transcription_loss(activations[:n], ground_truth, True))
    total = m['reconstruction'] * parts['reconstruction'] + m['transcription'] * parts['transcription']
    if m['consistency'] and trn_rec is not None:
        sp, sc = compute_consistency_loss(trn_rec[:n], trn_scr[:n], trn_coeffs[:n])
        parts['consistency_spectral'], parts['consistency_score'] = sp, sc
        total = total + m['consistency'] * (sp + sc)
    parts['total'] = total
    return total, parts
