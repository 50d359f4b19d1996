"""
Drop-in for reference ``timbre_trap/framework/objectives.py``: the three training objectives,
each one fused HIP reduction (forward) + one pointwise HIP kernel (backward) from csrc/losses.hip.
"""

from . import ops

__all__ = [
    'compute_reconstruction_loss',
    'compute_transcription_loss',
    'compute_consistency_loss'
]


def compute_reconstruction_loss(reconstructed, target):
    """
    Squared error summed over channel and frequency, averaged over batch and time
    (reference objectives.py:11-33).

    reconstructed, target : Tensor (B x C x F x T)  ->  scalar tensor
    """
    if reconstructed.shape != target.shape:
        raise ValueError('shape mismatch: %s vs %s' % (tuple(reconstructed.shape), tuple(target.shape)))
    # .sum(-3).sum(-2).mean(): every dim except (-3, -2) is averaged
    averaged = reconstructed.numel() // (reconstructed.size(-3) * reconstructed.size(-2))
    return ops.SqDiffLossFn.apply(reconstructed, target, 1.0 / averaged)


def compute_transcription_loss(estimate, target, weight_positive_class=False):
    """
    Squared error against the multi-pitch targets, optionally re-weighting exact positives by the
    per-frame negative/positive ratio (reference objectives.py:36-74).

    estimate, target : Tensor (B x F x T)  ->  scalar tensor
    """
    if estimate.shape != target.shape:
        raise ValueError('shape mismatch: %s vs %s' % (tuple(estimate.shape), tuple(target.shape)))
    return ops.TranscriptionLossFn.apply(estimate, target, bool(weight_positive_class))


def compute_consistency_loss(spectral_coefficients, transcription_coefficients, target):
    """
    Spectral- and score-consistency terms: two reconstruction losses against the (non-detached)
    transcription coefficients (reference objectives.py:77-104).
    """
    if spectral_coefficients.shape != target.shape or transcription_coefficients.shape != target.shape:
        raise ValueError('shape mismatch: %s, %s vs %s' % (tuple(spectral_coefficients.shape), tuple(transcription_coefficients.shape),
                                                           tuple(target.shape)))
    averaged = target.numel() // (target.size(-3) * target.size(-2))
    if target.numel() % 4 == 0 and all(t.dtype == target.dtype for t in (spectral_coefficients, transcription_coefficients)):
        # both terms share the target: one backward pass writes its gradient once (ops.SqDiff2Fn)
        return ops.SqDiff2Fn.apply(spectral_coefficients, transcription_coefficients, target, 1.0 / averaged)
    return (compute_reconstruction_loss(spectral_coefficients, target),
            compute_reconstruction_loss(transcription_coefficients, target))
