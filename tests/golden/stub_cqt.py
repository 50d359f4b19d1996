"""
A closed-form stand-in for the (absent) third-party ``cqt_pytorch.CQT`` used ONLY to drive the
reference's control flow when golden fixtures are generated (make_golden.py) and to replay the
same inputs through the oracle in tests.  It is our own code, not reference code, and it is not a
constant-Q transform: it just maps audio deterministically to (B,1,F,T) complex coefficients with
the right shapes (block_length samples -> max_window_length frames).
"""

import torch


def stub_encode(audio, n_bins, block_length, max_window_length):
    """audio (B,1,n*block_length) -> complex64 (B,1,n_bins,n*max_window_length)."""
    B = audio.size(0)
    hop = block_length // max_window_length
    n_frames = audio.size(-1) // hop
    s = audio.reshape(B, n_frames, hop).mean(-1)                      # (B, T)
    f = torch.arange(n_bins, dtype=audio.dtype).view(1, n_bins, 1)
    t = (torch.arange(n_frames) % max_window_length).to(audio.dtype).view(1, 1, n_frames)
    s = s.view(B, 1, n_frames)
    re = s * torch.cos(0.05 * f) + 0.10 * torch.cos(0.011 * f * (t + 1.0))
    im = s * torch.sin(0.03 * f + 0.2) + 0.10 * torch.sin(0.017 * f + 0.5 * t)
    return torch.complex(re, im).unsqueeze(1)


def closed_form_coefficients(B, n_bins, T, dtype=torch.float32):
    """Seed-free (B,2,n_bins,T) real coefficients for the autoencoder fixtures."""
    b = torch.arange(B, dtype=torch.float64).view(B, 1, 1, 1)
    c = torch.arange(2, dtype=torch.float64).view(1, 2, 1, 1)
    f = torch.arange(n_bins, dtype=torch.float64).view(1, 1, n_bins, 1)
    t = torch.arange(T, dtype=torch.float64).view(1, 1, 1, T)
    x = torch.sin(0.021 * f + 0.9 * t + 1.7 * c + 0.4 * b) + 0.5 * torch.cos(0.13 * f * (c + 1) - 0.3 * t)
    return x.to(dtype)


def closed_form_audio(B, n_samples, dtype=torch.float32):
    n = torch.arange(n_samples, dtype=torch.float64).view(1, 1, n_samples)
    b = torch.arange(B, dtype=torch.float64).view(B, 1, 1)
    x = 0.6 * torch.sin(0.05 * n + b) + 0.3 * torch.sin(0.31 * n + 0.5 * b * n / n_samples)
    return x.to(dtype)


def closed_form_targets(B, n_bins, T, dtype=torch.float32):
    """Blurred multi-pitch style targets in [0,1] with exact ones, and one all-zero frame."""
    f = torch.arange(n_bins, dtype=torch.float64).view(1, n_bins, 1)
    t = torch.arange(T, dtype=torch.float64).view(1, 1, T)
    b = torch.arange(B, dtype=torch.float64).view(B, 1, 1)
    centre = 40.0 + 37.0 * b + 11.0 * t
    centre2 = 300.0 - 23.0 * b + 5.0 * t
    g = torch.exp(-0.5 * (f - centre) ** 2) + torch.exp(-0.5 * (f - centre2) ** 2)
    g = g.clamp(0, 1)
    g[:, :, 0] = 0.0          # a frame without positives -> eps path
    return g.to(dtype)
