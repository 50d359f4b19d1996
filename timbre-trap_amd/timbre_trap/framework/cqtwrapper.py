"""
Drop-in for reference ``timbre_trap/framework/cqtwrapper.py`` (class ``CQT``): same constructor,
attributes (``sample_rate, hop_length, n_bins, midi_freqs, block_length, max_window_length``) and
methods, with the transform itself (``cqt_pytorch.CQT.encode/decode`` in the reference) computed by
the gfx950 kernels of csrc/cqt.hip through the C ABI ``tt_cqt_forward`` / ``tt_cqt_inverse``.

The frame/time helpers (reference :215-308) stay pure Python/NumPy: the datasets call them from
DataLoader worker processes, so they must not touch the HIP runtime, and the object stays
fork- and pickle-safe (no library handle is stored on it).
"""

import ctypes
import math

import numpy as np
import torch
import torch.nn as nn

from .. import _hip
from . import nsgt_plan

_TABLES = ('tw675', 'tw49', 'twNc', 'twN', 'tw1024', 'bin_tab', 'window', 'dual', 'gat_off', 'gat_idx')
_GENERIC_TABLES = ('chirp', 'bfilt', 'twP', 'twM', 'bin_tab', 'window', 'dual', 'gat_off', 'gat_idx', 'pos_bin')
# True: also the reference configuration (3 s @ 22.05 kHz) runs on the any-block-length kernels (csrc/cqt_generic.hip) -- for tests
# that hold the two paths against each other; read when a CQT is constructed
FORCE_GENERIC = False


class CQT(nn.Module):
    """
    Invertible constant-Q transform (NSGT, one block of ``secs_per_block`` seconds at a time).
    """

    def __init__(self, n_octaves, bins_per_octave, sample_rate, secs_per_block, conventions=None):
        """
        Parameters (reference cqtwrapper.py:15-29; ``conventions`` is an addition: an
        ``nsgt_plan.NSGTConventions`` selecting the window family / rounding / crop alignment / dual-window rule that the
        reference leaves to ``cqt_pytorch`` -- table-level switches, default documented in INTEGRATION.md)
        ----------
        n_octaves : int
          Number of octaves below Nyquist to span
        bins_per_octave : int
          Number of bins allocated to each octave
        sample_rate : int or float
          Number of samples per second of audio
        secs_per_block : float
          Number of seconds to process at a time
        """
        super().__init__()

        self.conventions = nsgt_plan.DEFAULT_CONVENTIONS if conventions is None else conventions
        plan = nsgt_plan.build_plan(n_octaves, bins_per_octave, sample_rate,
                                    int(secs_per_block * sample_rate), power_of_2_length=True, conventions=self.conventions,
                                    generic=FORCE_GENERIC)
        self.block_length = plan['N']
        self.max_window_length = plan['M']
        self._sum_len = plan['sum_len']
        # the reference configuration (N = 66150, M = 1024) runs on the specialised kernels of csrc/cqt.hip, every other block
        # length on the any-length path of csrc/cqt_generic.hip (same tables, same conventions; reference cqtwrapper.py:15-48
        # accepts any secs_per_block / sample_rate)
        self._fast = plan['fast']
        self._P = int(plan.get('P', 0))

        for name in (_TABLES if self._fast else _GENERIC_TABLES):
            a = plan[name]
            t = torch.from_numpy(np.ascontiguousarray(a)).to(torch.int32 if a.dtype.kind == 'i' else torch.float32)
            self.register_buffer('_t_' + name, t.contiguous(), persistent=False)

        self.sample_rate = sample_rate
        # reference cqtwrapper.py:40-48
        self.hop_length = (self.block_length / self.max_window_length)
        self.n_bins = n_octaves * bins_per_octave
        fmin = 12 * (np.log2(np.asanyarray((sample_rate / 2) / (2 ** n_octaves))) - np.log2(440.0)) + 69
        self.midi_freqs = fmin + np.arange(self.n_bins) / (bins_per_octave / 12)

    # ---- device side -------------------------------------------------------------------------

    def _plan_struct(self, device):
        if self._t_window.device != device:
            raise RuntimeError('CQT tables are on %s but the input is on %s; call .to(device) on the module'
                               % (self._t_window.device, device))
        ps = _hip.CqtPlan() if self._fast else _hip.CqtGenericPlan()
        for name in (_TABLES if self._fast else _GENERIC_TABLES):
            setattr(ps, name, getattr(self, '_t_' + name).data_ptr())
        ps.n_bins = self.n_bins
        ps.sum_len = self._sum_len
        if not self._fast:
            ps.N, ps.M, ps.P = self.block_length, self.max_window_length, self._P
        return ps

    def _scratch(self, lib, ps, n_clips, device):
        nbytes = (lib.tt_cqt_scratch_bytes(n_clips, self.n_bins, self._sum_len) if self._fast
                  else lib.tt_cqt_generic_scratch_bytes(ctypes.byref(ps), n_clips))
        if nbytes <= 0:
            raise RuntimeError('the constant-Q transform plan was rejected by the library (block_length=%d, max_window_length=%d)'
                               % (self.block_length, self.max_window_length))
        return torch.empty(nbytes, dtype=torch.uint8, device=device)

    def _prepare_audio(self, audio):
        _hip.require_cuda(audio)
        if audio.size(-1) % self.block_length:
            raise ValueError('audio length %d is not a multiple of the block length %d (use pad_to_block_length)'
                             % (audio.size(-1), self.block_length))
        lead = audio.shape[:-1]
        a = audio.reshape(-1, audio.size(-1)).to(torch.float32).contiguous()
        return a, lead

    def _transform(self, audio, complex_out):
        a, lead = self._prepare_audio(audio)
        B, n_blocks = a.size(0), a.size(1) // self.block_length
        T = n_blocks * self.max_window_length
        lib = _hip.lib()
        ps = self._plan_struct(a.device)
        scratch = self._scratch(lib, ps, B * n_blocks, a.device)
        forward = lib.tt_cqt_forward if self._fast else lib.tt_cqt_generic_forward
        if complex_out:
            out = torch.empty((B, 1, self.n_bins, T, 2), dtype=torch.float32, device=a.device)
        else:
            out = torch.empty((B, 2, self.n_bins, T), dtype=torch.float32, device=a.device)
        with _hip.timed('cqt_forward'):
            _hip.check(forward(ctypes.byref(ps), _hip.ptr(a), _hip.ptr(out), _hip.ptr(scratch),
                               B, n_blocks, int(complex_out), _hip.stream_ptr()), 'tt_cqt_forward')
        if complex_out:
            out = torch.view_as_complex(out)
        # (B,1,N) input -> lead == (B,1): the channel dim of the reference output replaces it
        if len(lead) != 2:
            out = out.reshape(*lead, *out.shape[1:])
        return out

    def encode(self, audio):
        """audio (B x 1 x N) -> complex coefficients (B x 1 x F x T)   (cqtwrapper.py:67, sonify.py:94)."""
        with torch.no_grad():
            return self._transform(audio, True)

    def forward(self, audio):
        """
        Encode a batch of audio into CQT spectral coefficients (reference cqtwrapper.py:50-72).

        audio : Tensor (B x 1 x T)  ->  coefficients : Tensor (B x 2 x F x T) real/imaginary
        """
        with torch.no_grad():
            return self._transform(audio, False)

    def decode(self, coefficients):
        """
        Invert CQT spectral coefficients to synthesize audio (reference cqtwrapper.py:184-213).

        coefficients : Tensor (B x 2 OR 1 x F x T) real/imaginary OR complex -> audio (B x 1 x T),
        divided by its infinity norm when that is non-zero.
        """
        return self._decode(coefficients, True)

    def _decode_raw(self, coefficients):
        """The synthesis without the batch-wide division by the infinity norm (what ``cqt_pytorch.CQT.decode`` returns before
        reference cqtwrapper.py:209-211 rescales it); used by tests to compare clips of one batch independently."""
        return self._decode(coefficients, False)

    def _decode(self, coefficients, normalize):
        with torch.no_grad():
            _hip.require_cuda(coefficients)
            is_complex = coefficients.is_complex()
            if is_complex:
                c = torch.view_as_real(coefficients.to(torch.complex64).contiguous())
                B = c.size(0)
                T = c.size(-2)
            else:
                c = coefficients.to(torch.float32).contiguous()
                B = c.size(0)
                T = c.size(-1)
            if T % self.max_window_length:
                raise ValueError('frame count %d is not a multiple of %d' % (T, self.max_window_length))
            n_blocks = T // self.max_window_length
            lib = _hip.lib()
            ps = self._plan_struct(c.device)
            scratch = self._scratch(lib, ps, B * n_blocks, c.device)
            inverse = lib.tt_cqt_inverse if self._fast else lib.tt_cqt_generic_inverse
            audio = torch.empty((B, 1, n_blocks * self.block_length), dtype=torch.float32, device=c.device)
            with _hip.timed('cqt_inverse'):
                _hip.check(inverse(ctypes.byref(ps), _hip.ptr(c), _hip.ptr(audio), _hip.ptr(scratch),
                                   B, n_blocks, int(is_complex), int(bool(normalize)), _hip.stream_ptr()), 'tt_cqt_inverse')
        return audio

    # ---- layout helpers (reference cqtwrapper.py:74-182), stock tensor views ----------------------

    @staticmethod
    def to_real(coefficients):
        """complex (B x 1 x F x T) -> real/imaginary (B x 2 x F x T)   (cqtwrapper.py:74-97)."""
        coefficients = coefficients.squeeze(-3)
        coefficients = torch.view_as_real(coefficients)
        return coefficients.transpose(-1, -2).transpose(-2, -3)

    @staticmethod
    def to_complex(coefficients):
        """real/imaginary (B x 2 x F x T) -> complex (B x F x T)   (cqtwrapper.py:99-120)."""
        coefficients = coefficients.transpose(-3, -2).transpose(-2, -1)
        return torch.view_as_complex(coefficients.contiguous())

    @staticmethod
    def to_magnitude(coefficients):
        """L2 norm over the real/imaginary channel (cqtwrapper.py:122-141)."""
        return coefficients.norm(p=2, dim=-3)

    @staticmethod
    def to_decibels(magnitude, rescale=True):
        """
        Amplitude -> dB per track with an 80 dB floor, optionally rescaled to [0, 1]
        (cqtwrapper.py:143-182; torchaudio AmplitudeToDB('amplitude', top_db=80) restated).
        """
        decibels = list()
        for m in magnitude:
            d = 20.0 * torch.log10(torch.clamp(m, min=1e-10))
            d = torch.maximum(d, d.max() - 80.0)
            if rescale:
                d = d - d.max()
                d = 1 + d / 80
            decibels.append(d.unsqueeze(0))
        return torch.cat(decibels, dim=0)

    # ---- frame / time arithmetic (pure Python; reference cqtwrapper.py:215-308) -------------------

    def pad_to_block_length(self, audio):
        """Zero-pad to the next multiple of the block length (cqtwrapper.py:215-233)."""
        return torch.nn.functional.pad(audio, (0, -audio.size(-1) % self.block_length))

    def get_expected_samples(self, t):
        """Number of samples for ``t`` seconds, rounded down (cqtwrapper.py:235-253)."""
        return int(max(0, t) * self.sample_rate)

    def get_expected_frames(self, num_samples):
        """Number of frames returned for ``num_samples`` samples (cqtwrapper.py:255-273)."""
        return math.ceil((num_samples / self.block_length) * self.max_window_length)

    def get_times(self, n_frames):
        """Time in seconds of each frame (cqtwrapper.py:275-293)."""
        return np.arange(n_frames) * self.hop_length / self.sample_rate

    def get_midi_freqs(self):
        """Centre frequency (MIDI) of each bin (cqtwrapper.py:295-308)."""
        return self.midi_freqs

    # reference checkpoints carry cqt_pytorch buffers under sliCQ.*; ours are derived, so ignore them
    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        for k in [k for k in state_dict if k.startswith(prefix)]:
            state_dict.pop(k)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)
