"""
timbre_trap -- MI355X-native drop-in for the hot path of sony/timbre-trap
(``timbre_trap.framework``: CQT, TimbreTrap, objectives).  Device work is done by
hand-written gfx950 kernels in libttrap_hip.so (timbre-trap_amd/csrc, C ABI in include/ttrap.h).

Overlay: the reference's ``timbre_trap`` directory is a namespace package (it has no ``__init__.py``) whose sub-packages the
reference scripts import directly (``timbre_trap.datasets.MixedMultiPitch``, ``timbre_trap.utils.data`` ...).  This package
replaces ``timbre_trap.framework`` entirely and the path-adjacent parts of ``timbre_trap.utils``; everything it does NOT
provide (dataset wrappers, download helpers, plotting) falls through to a reference checkout found later on ``sys.path``:
``__path__`` of this package, of ``.utils`` and of ``.datasets`` is extended with the matching reference directories.
With no reference on the path nothing changes.
"""

from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
__version__ = '0.2.0'
