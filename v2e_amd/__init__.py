"""v2e_amd -- the v2e hot path (DVS event emulator + SuperSloMo interpolation) on MI355X.

`EventEmulator` and `SuperSloMo` keep the reference's class signatures
(v2ecore/emulator.py:86-117, v2ecore/slomo.py:44-54) so that v2e.py can use them as
drop-in replacements.  All arithmetic runs in hand-written HIP
kernels for gfx950 (v2e_amd/csrc, C ABI in include/v2e_amd.h); there is no CPU path.
"""
from .emulator import EventEmulator  # noqa: F401
from .slomo import SuperSloMo  # noqa: F401
from .renderer import EventRenderer, ExposureMode  # noqa: F401
from .preproc import Stage1  # noqa: F401  (v2e.py stage 1: INTER_AREA resize + BGR2GRAY on device; parity unpinned)
from ._capi import V2EAmdError  # noqa: F401

__all__ = ["EventEmulator", "SuperSloMo", "EventRenderer", "ExposureMode", "Stage1", "V2EAmdError"]
