"""
timbre_trap -- MI355X-native drop-in for the hot path of sony/timbre-trap
(``timbre_trap.framework``: CQT, TimbreTrap, objectives).  Device work is done by
hand-written gfx950 kernels in libttrap_hip.so (timbre-trap_amd/csrc, C ABI in include/ttrap.h).
"""

__version__ = '0.1.0'
