from .optim import FusedAdamW
from .distributed import DataParallel, GradientSync, init_process_group_from_env, allreduce_gradients
from .experiments import (CosineWarmup, seed_everything, gradient_statistics, sum_gradient_norms, average_gradient_norms,
                          get_max_gradient, get_max_gradient_norm)
from .processing import to_array, filter_non_peaks, threshold, peaks_above
from .targets import multi_pitch_to_activations, activations_to_multi_pitch, hz_to_midi, midi_to_hz
