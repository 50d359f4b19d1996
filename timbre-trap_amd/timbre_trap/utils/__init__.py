"""
``timbre_trap.utils``: the path-adjacent helpers of the reference package, implemented here (HIP kernels where the work is
device work), plus -- when a reference checkout is on ``sys.path`` after this package -- the reference's own host-only modules
this package does not replace (``data``: constants / download helpers, ``visualization``), re-exported under the reference's
names so that ``from timbre_trap.utils import *`` yields the same namespace as in the reference.
"""

from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)

from .optim import FusedAdamW
from .distributed import DataParallel, GradientSync, init_process_group_from_env, allreduce_gradients
from .experiments import (CosineWarmup, seed_everything, gradient_statistics, sum_gradient_norms, average_gradient_norms,
                          get_max_gradient, get_max_gradient_norm)
from .processing import to_array, debug_nans, filter_non_peaks, threshold, peaks_above
from .targets import multi_pitch_to_activations, activations_to_multi_pitch, hz_to_midi, midi_to_hz
from .slicing import slice_audio, slice_times, resample_multi_pitch, nearest_indices, ExcerptSlicer
from .metrics import MultipitchEvaluator, multipitch_metrics, signal_distortion_ratio
from .trainloop import (make_schedulers, checkpoints_for, StepLogger, TrainingState, save_checkpoint, print_and_log,
                        log_gradient_norms, TRAIN_TAGS)

# reference-only host modules (found through the extended __path__; absent or missing a third-party dependency -> skipped)
REFERENCE_MODULES = {}
for _name in ('data', 'visualization'):
    try:
        _mod = __import__(__name__ + '.' + _name, fromlist=['*'])
    except Exception as _e:                      # ImportError, or a dependency of the reference module failing to import
        REFERENCE_MODULES[_name] = repr(_e)
        continue
    REFERENCE_MODULES[_name] = _mod.__file__
    for _k in getattr(_mod, '__all__', ()):
        globals()[_k] = getattr(_mod, _k)
del _name
