"""
``timbre_trap.datasets``.  File parsing, downloads and the dataset classes are outside the accelerated path (DESIGN.md
section 6): when a reference checkout is on ``sys.path`` after this package, its dataset modules are used as they are -- this
package's ``__path__`` is extended with the reference directory, so ``timbre_trap.datasets.MixedMultiPitch`` etc. resolve
there and the base classes below are the reference's.  Without a reference checkout only the path-adjacent pieces exist
(SURVEY.md section 8f, rows f3 / f4): the target helpers as static methods and the slicing arithmetic.
"""

from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)

from ..utils.slicing import ExcerptSlicer
from ..utils.targets import activations_to_multi_pitch, multi_pitch_to_activations

REFERENCE_DATASETS = None
try:
    from .BaseDataset import BaseDataset, ComboDataset, StemMixingDataset
    from .AudioDataset import AudioDataset
    from .PitchDataset import PitchDataset
    from .MPEDataset import MPEDataset
    from .NoteDataset import NoteDataset
    from .AMTDataset import AMTDataset
    REFERENCE_DATASETS = BaseDataset.__module__
except Exception as _e:                          # no reference on the path, or one of its third-party imports is missing
    REFERENCE_DATASETS = None
    _reason = repr(_e)

    class PitchDataset(ExcerptSlicer):
        """
        Stand-in with the reference's static target helpers (``PitchDataset.multi_pitch_to_activations`` /
        ``activations_to_multi_pitch``, device kernels) and slicing methods (``slice_times``, ``resample_multi_pitch``);
        construct it with the CQT object and excerpt length instead of a dataset directory.
        """

        multi_pitch_to_activations = staticmethod(multi_pitch_to_activations)
        activations_to_multi_pitch = staticmethod(activations_to_multi_pitch)

    class AudioDataset(ExcerptSlicer):
        """Stand-in with the reference's ``slice_audio`` arithmetic."""
