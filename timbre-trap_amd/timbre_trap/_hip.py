"""
ctypes binding of libttrap_hip.so (C ABI: include/ttrap.h).

Loaded lazily on first use so that importing the package -- and pickling / forking a CQT object
into DataLoader workers, which only call the pure-Python helpers (reference
timbre_trap/datasets/PitchDataset.py:102-118) -- never touches the HIP runtime.

There is NO CPU fallback: if the library is missing, every device entry point raises.
"""

import ctypes
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)                       # timbre-trap_amd/
REPO_ROOT = os.path.dirname(PKG_ROOT)
CSRC = os.path.join(PKG_ROOT, 'csrc')
LIB_PATH = os.path.join(PKG_ROOT, 'lib', 'libttrap_hip.so')
if os.environ.get('TTRAP_LIB'):                         # tuning: an alternative build of the same sources (tools/build_variant.sh)
    LIB_PATH = os.path.join(PKG_ROOT, 'lib', os.environ['TTRAP_LIB'])
SOURCES = ['cqt.hip', 'cqt_generic.hip', 'conv_generic.hip', 'conv_mfma.hip', 'conv_small.hip', 'conv_wide_bf16.hip', 'conv_level_bf16.hip', 'conv_stride_bf16.hip',
           'latent_bf16.hip', 'conv_edge_bf16.hip', 'gemm.hip', 'losses.hip',
           # the 16-bit channels-last sources a second time with fp16 elements (two-line wrappers: #define TT_F16 + #include)
           'conv_wide_f16.hip', 'conv_level_f16.hip', 'conv_stride_f16.hip', 'latent_f16.hip', 'conv_edge_f16.hip',
           # fp32-class inference blocks on split fp16 operands
           'conv_x3.hip']

_lib = None

c_void_p, c_int, c_int64, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float


class CqtPlan(ctypes.Structure):
    """struct tt_cqt_plan (include/ttrap.h)."""
    _fields_ = [('tw675', c_void_p), ('tw49', c_void_p), ('twNc', c_void_p), ('twN', c_void_p),
                ('tw1024', c_void_p), ('bin_tab', c_void_p), ('window', c_void_p), ('dual', c_void_p),
                ('gat_off', c_void_p), ('gat_idx', c_void_p), ('n_bins', ctypes.c_int32),
                ('sum_len', ctypes.c_int32)]


class CqtGenericPlan(ctypes.Structure):
    """struct tt_cqt_gplan (include/ttrap.h): the any-block-length transform of csrc/cqt_generic.hip."""
    _fields_ = [('chirp', c_void_p), ('bfilt', c_void_p), ('twP', c_void_p), ('twM', c_void_p), ('bin_tab', c_void_p), ('window', c_void_p),
                ('dual', c_void_p), ('gat_off', c_void_p), ('gat_idx', c_void_p), ('pos_bin', c_void_p), ('n_bins', ctypes.c_int32),
                ('sum_len', ctypes.c_int32), ('N', ctypes.c_int32), ('M', ctypes.c_int32), ('P', ctypes.c_int32)]


P, I, L, F_ = c_void_p, c_int, c_int64, c_float
_PROTOS = {
    'tt_version': (c_int, []),
    'tt_arch': (ctypes.c_char_p, []),
    'tt_error_string': (ctypes.c_char_p, [I]),
    'tt_set_cu_limit': (c_int, [I]),
    'tt_cqt_scratch_bytes': (c_int64, [I, I, I]),
    'tt_cqt_forward': (c_int, [ctypes.POINTER(CqtPlan), P, P, P, I, I, I, P]),
    'tt_cqt_inverse': (c_int, [ctypes.POINTER(CqtPlan), P, P, P, I, I, I, I, P]),
    'tt_cqt_generic_scratch_bytes': (c_int64, [ctypes.POINTER(CqtGenericPlan), I]),
    'tt_cqt_generic_forward': (c_int, [ctypes.POINTER(CqtGenericPlan), P, P, P, I, I, I, P]),
    'tt_cqt_generic_inverse': (c_int, [ctypes.POINTER(CqtGenericPlan), P, P, P, I, I, I, I, P]),
    'tt_conv2d': (c_int, [P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, I, L, L, L, L, I, P]),
    'tt_conv2d_wgrad': (c_int, [P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, L, L, L, L, P]),
    'tt_elu_bwd': (c_int, [P, P, P, L, P]),
    'tt_resblock_fwd': (c_int, [P, P, P, P, P, P, P, I, I, I, I, I, I, P]),
    'tt_resblock_bwd': (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, P]),
    'tt_wide_scratch_bytes': (c_int64, [I, I, I, I]),
    'tt_wide_pack': (c_int, [P, P, I, I, I, I, P]),
    'tt_wide_unpack': (c_int, [P, P, I, I, I, I, P]),
    'tt_wide_rb_fwd': (c_int, [P, P, P, P, P, P, P, I, I, I, I, I, P]),
    'tt_wide_rb_fwd_join': (c_int, [P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, P]),
    'tt_wide_rb_bwd': (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, P]),
    'tt_wide_level_scratch_bytes': (c_int64, [I, I, I, I, I]),
    'tt_wide_level_bwd': (c_int, [I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, P, P]),
    'tt_wide_level_bwd_gated': (c_int, [I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, P, P]),
    'tt_wide_level_bwd_gated_join': (c_int, [I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, P, P, I, P, I, P, P]),
    'tt_x3_bytes': (c_int64, [I, I, I, I]),
    'tt_x3_level_scratch_bytes': (c_int64, [I, I, I, I]),
    'tt_x3_pack': (c_int, [P, P, I, I, I, I, P]),
    'tt_x3_unpack': (c_int, [P, P, I, I, I, I, P]),
    'tt_x3_rb_fwd': (c_int, [P, P, P, P, P, P, I, I, I, I, I, I, P]),
    'tt_x3_level_fwd': (c_int, [I, P, I, P, I, P, P, P, P, P, P, I, I, I, I, P]),
    'tt_x3n_level_scratch_bytes': (c_int64, [I, I, I, I]),
    'tt_x3n_rb_fwd': (c_int, [P, I, P, P, P, P, P, I, I, I, I, I, I, P]),
    'tt_x3n_level_fwd': (c_int, [I, P, P, P, P, P, P, P, P, I, I, I, I, P]),
    'tt_x3_latent_scratch_bytes': (c_int64, [I, I, I]),
    'tt_x3_latent_encode': (c_int, [P, P, P, P, P, I, I, I, I, I, P]),
    'tt_x3_latent_decode': (c_int, [P, I, c_float, P, P, P, I, P, I, I, I, I, I, P]),
    'tt_x3_sconv_fwd': (c_int, [P, I, P, P, P, I, I, I, I, I, P]),
    'tt_x3_tconv_fwd': (c_int, [P, I, P, P, P, I, I, I, I, I, I, P]),
    'tt_wide_fused_scratch_bytes': (c_int64, [I]),
    'tt_wide_rb_bwd_fused': (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, P]),
    'tt_wide_onepass_scratch_bytes': (c_int64, [I]),
    'tt_wide_rb_bwd_onepass': (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, P]),
    'tt_wide_rb_bwd_is_onepass': (c_int, [I, I]),
    'tt_stride16_scratch_bytes': (c_int64, [I]),
    'tt_sconv16_fwd': (c_int, [P, P, P, P, I, I, I, I, P]),
    'tt_sconv16_bwd': (c_int, [P, P, P, P, P, P, P, P, I, I, I, I, P]),
    'tt_tconv16_fwd': (c_int, [P, P, P, P, I, I, I, I, I, P]),
    'tt_tconv16_bwd': (c_int, [P, P, P, P, P, P, P, P, I, I, I, I, I, P]),
    'tt_gate16': (c_int, [P, P, L, P]),
    'tt_sconv16_bwd_pregated': (c_int, [P, P, P, P, P, P, P, I, I, I, I, P]),
    'tt_tconv16_bwd_pregated': (c_int, [P, P, P, P, P, P, P, I, I, I, I, I, I, P]),
    'tt_latent16_pregated_ok': (c_int, [I, I]),
    'tt_latent16_expand_gated': (c_int, [P, P, P, P, P, I, I, I, I, I, P]),
    'tt_latent16_contract_pregated': (c_int, [P, P, P, P, I, I, I, I, I, I, P]),
    'tt_latent16_wgrad_pregated': (c_int, [P, I, F_, P, P, P, P, I, I, I, I, I, P]),
    'tt_latent16_scratch_bytes': (c_int64, [I, I, I, I, I]),
    'tt_latent16_contract': (c_int, [P, P, P, P, P, P, I, I, I, I, I, I, P]),
    'tt_latent16_expand': (c_int, [P, I, F_, P, P, P, P, I, I, I, I, I, P]),
    'tt_latent16_wgrad': (c_int, [P, I, F_, P, P, P, P, P, I, I, I, I, I, P]),
    'tt_edge16_scratch_bytes': (c_int64, []),
    'tt_convin16_fwd': (c_int, [P, P, P, P, I, I, I, P]),
    'tt_convin16_bwd': (c_int, [P, P, P, P, P, P, P, P, I, I, I, P]),
    'tt_convout16_fwd': (c_int, [P, P, P, P, I, I, I, P]),
    'tt_convout16_bwd': (c_int, [P, P, P, P, P, P, P, I, I, I, P]),
    'tt_sconv_fwd': (c_int, [P, P, P, P, I, I, I, I, P]),
    'tt_wgrad_scratch_floats': (c_int64, []),
    'tt_sconv_bwd': (c_int, [P, P, P, P, P, P, P, P, I, I, I, I, P]),
    'tt_tconv_fwd': (c_int, [P, P, P, P, I, I, I, I, I, P]),
    'tt_tconv_bwd': (c_int, [P, P, P, P, P, P, P, P, I, I, I, I, I, P]),
    'tt_gemm': (c_int, [P, P, P, P, I, I, I, I, I, L, L, L, I, L, L, L, I, F_, F_, I, I, I, P]),
    'tt_channel_sum': (c_int, [P, P, I, I, L, P]),
    'tt_scaled_add': (c_int, [P, P, P, I, P, L, P]),
    'tt_window_ola': (c_int, [P, P, P, L, I, I, I, L, P]),
    'tt_dot': (c_int, [P, P, P, L, P]),
    'tt_scaled_add16': (c_int, [P, P, P, I, P, L, P]),
    'tt_dot16': (c_int, [P, P, P, L, P]),
    'tt_skip_join16_fwd': (c_int, [P, P, P, I, P, L, I, P]),
    'tt_skip_join16_bwd': (c_int, [P, P, P, I, P, P, L, I, I, P]),
    'tt_sqdiff_sum': (c_int, [P, P, P, P, L, F_, P]),
    'tt_sqdiff_bwd': (c_int, [P, P, P, F_, P, P, L, P]),
    'tt_sqdiff2_bwd': (c_int, [P, P, P, P, P, F_, P, P, P, L, P]),
    'tt_sqdiff_sum_grad': (c_int, [P, P, P, P, L, F_, P, P, P]),
    'tt_sqdiff2_sum_grad': (c_int, [P, P, P, P, P, P, L, F_, P, P, P, P]),
    'tt_sqdiff_rescale': (c_int, [P, P, P, L, P]),
    'tt_sqdiff2_rescale': (c_int, [P, P, P, P, P, L, P]),
    'tt_activations_fwd': (c_int, [P, P, I, I, I, P]),
    'tt_activations_bwd': (c_int, [P, P, P, P, I, I, I, P]),
    'tt_transcription_loss_fwd': (c_int, [P, P, P, P, P, I, I, I, I, P]),
    'tt_transcription_loss_fwd_grad': (c_int, [P, P, P, P, P, P, I, I, I, I, P]),
    'tt_transcription_loss_bwd': (c_int, [P, P, P, P, P, I, I, I, I, P]),
    'tt_segment_stats': (c_int, [P, P, I, P, P]),
    'tt_peak_pick': (c_int, [P, P, L, I, I, ctypes.c_double, I, I, P]),
    'tt_target_activations': (c_int, [P, P, I, P, I, I, I, P, P, P]),
    'tt_l2norm': (c_int, [P, P, P, L, P]),
    'tt_adamw_step': (c_int, [P, P, P, P, P, L, F_, F_, F_, F_, F_, I, F_, I, P, P]),
    'tt_set_loss_scale': (c_float, [F_]),
}
# fp16 twins (include/ttrap.h: suffix _h): the 16-bit channels-last sources compiled a second time with -DTT_F16
HALF_TWINS = ('tt_wide_level_scratch_bytes', 'tt_wide_level_bwd', 'tt_wide_level_bwd_gated', 'tt_wide_level_bwd_gated_join', 'tt_gate16', 'tt_latent16_pregated_ok', 'tt_latent16_expand_gated', 'tt_latent16_contract_pregated', 'tt_latent16_wgrad_pregated', 'tt_sconv16_bwd_pregated', 'tt_tconv16_bwd_pregated', 'tt_wide_scratch_bytes', 'tt_wide_pack', 'tt_wide_unpack', 'tt_wide_rb_fwd', 'tt_wide_rb_fwd_join', 'tt_wide_rb_bwd', 'tt_wide_fused_scratch_bytes',
              'tt_wide_rb_bwd_fused', 'tt_wide_onepass_scratch_bytes', 'tt_wide_rb_bwd_onepass', 'tt_wide_rb_bwd_is_onepass',
              'tt_stride16_scratch_bytes', 'tt_sconv16_fwd', 'tt_sconv16_bwd', 'tt_tconv16_fwd', 'tt_tconv16_bwd', 'tt_latent16_scratch_bytes',
              'tt_latent16_contract', 'tt_latent16_expand', 'tt_latent16_wgrad', 'tt_edge16_scratch_bytes', 'tt_convin16_fwd', 'tt_convin16_bwd',
              'tt_convout16_fwd', 'tt_convout16_bwd', 'tt_scaled_add16', 'tt_dot16', 'tt_skip_join16_fwd', 'tt_skip_join16_bwd')
for _n in HALF_TWINS:
    _PROTOS[_n + '_h'] = _PROTOS[_n]
EXPORTED_SYMBOLS = tuple(_PROTOS)
# Per-source compiler flags.  cqt.hip: the SLP vectoriser turns the complex butterflies into v_pk_*_f32 plus the register moves that
# form their aligned pairs; on gfx950 a packed fp32 op issues at about half the rate of a plain one (tools/probes/pkfma_probe.cpp:
# 5.1 vs 2.8 cycles), so the pairs buy nothing and the moves cost: CQT forward 0.122 -> 0.109 ms, inverse 0.175 -> 0.161 ms without it.
# The other sources keep it (it also merges loads / stores: the residual backward is 0.5 ms per step slower without).
EXTRA_FLAGS = {'cqt.hip': ['-fno-slp-vectorize']}


def build(verbose=False, force=False):
    """Compile every HIP source for gfx950 into timbre-trap_amd/lib/libttrap_hip.so (in-tree)."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, 'common.h'), os.path.join(REPO_ROOT, 'include', 'ttrap.h')]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')] + [os.path.abspath(__file__)]
    if os.environ.get('TTRAP_LIB'):
        # an alternative build selected for an A/B run (tools/build_variant.sh made it with ITS flags): never rebuilt from here --
        # that would overwrite the variant with the default flags and compare default against default
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('TTRAP_LIB=%s: %s does not exist (build it with tools/build_variant.sh)' % (os.environ['TTRAP_LIB'], LIB_PATH))
        return LIB_PATH
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return LIB_PATH
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    jobs = []
    for s in srcs:
        base = os.path.basename(s)
        o = os.path.join(PKG_ROOT, 'lib', base.replace('.hip', '.o'))
        objs.append(o)
        jobs.append([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC'] + EXTRA_FLAGS.get(base, []) + ['-c', s, '-o', o])
    # at most as many compilers at once as there are cores (the largest sources take ~1 GB each)
    width = max(1, min(len(jobs), os.cpu_count() or 1))
    running, failed = [], None
    pending = list(jobs)
    while pending or running:
        while pending and len(running) < width:
            cmd = pending.pop(0)
            if verbose:
                print(' '.join(cmd), file=sys.stderr)
            running.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        cmd, pr = running.pop(0)
        out, _ = pr.communicate()
        if pr.returncode != 0 and failed is None:
            failed = 'hipcc failed: %s\n%s' % (' '.join(cmd), out.decode())
            pending = []
    if failed:
        raise RuntimeError(failed)
    # objects of sources that no longer exist must not travel to the GPU box with the snapshot
    for f in os.listdir(os.path.dirname(LIB_PATH)):
        if f.endswith('.o') and os.path.join(os.path.dirname(LIB_PATH), f) not in objs:
            os.remove(os.path.join(os.path.dirname(LIB_PATH), f))
    # link to a temporary name and rename into place: another rank must never dlopen a half-written library
    tmp = LIB_PATH + '.tmp.%d' % os.getpid()
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', tmp] + objs
    subprocess.run(cmd, check=True)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


def lib():
    """The loaded library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'libttrap_hip.so not found at %s -- run `python -c "import __graft_entry__ as g; g.build()"` '
                '(there is no CPU fallback for the Timbre-Trap HIP path)' % LIB_PATH)
        # torch bundles its own HIP runtime: load it FIRST so that libttrap_hip.so's libamdhip64.so.7 resolves
        # to the same copy (two HIP runtimes in one process cannot both see the device)
        import torch  # noqa: F401
        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(h, name)          # AttributeError if a declared symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = lib().tt_error_string(rc)
        raise RuntimeError('ttrap HIP call %s failed (%d): %s' % (what, rc, msg.decode() if msg else '?'))


# bench.py sets this to a dict to time C-ABI calls with HIP events recorded on the launch stream; EVENT_KEYS (a set or
# None = every instrumented call) restricts which calls are bracketed
EVENT_LOG = None
EVENT_KEYS = None


class timed:
    """with timed(key): <C-ABI call>  -- records a (start, end, clips) event triple when EVENT_LOG is a dict; ``clips`` = the batch of
    that call (None: unknown) -- the train step runs its decoder over 2 B clips per call and its encoder over B, and bench.py prices a call
    per clip."""

    def __init__(self, key, clips=None):
        self.key = key
        self.clips = clips
        self.start = None

    def __enter__(self):
        if EVENT_LOG is not None and (EVENT_KEYS is None or self.key in EVENT_KEYS):
            import torch
            self.start = torch.cuda.Event(enable_timing=True)
            self.start.record()
        return self

    def __exit__(self, *exc):
        if self.start is not None:
            import torch
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            EVENT_LOG.setdefault(self.key, []).append((self.start, end, self.clips))
        return False


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """data pointer of a torch tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('timbre_trap HIP path needs tensors on the GPU (got %s); there is no CPU fallback'
                               % t.device)
