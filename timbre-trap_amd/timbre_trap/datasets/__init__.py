"""
Only the two path-adjacent static methods of the reference's ``PitchDataset`` live here;
file parsing, slicing and the dataset classes themselves are out of scope.
"""

from ..utils.targets import activations_to_multi_pitch, multi_pitch_to_activations


class PitchDataset:
    """Namespace with the reference's call signatures: ``PitchDataset.multi_pitch_to_activations(...)`` etc."""

    multi_pitch_to_activations = staticmethod(multi_pitch_to_activations)
    activations_to_multi_pitch = staticmethod(activations_to_multi_pitch)

    def __init__(self, *args, **kwargs):
        raise NotImplementedError('dataset loading is outside the accelerated path; only the static target helpers exist here')
