"""Launcher that runs the reference's own `v2e.py` with the MI355X hot path bound in.

`v2e.py:38-39` binds `SuperSloMo` and `EventEmulator` as module globals at import time; nothing in the reference tree
is edited: this module imports `v2e` from a checkout, rebinds the two names (there and in the modules they came from, for
`v2ecore/renderer.py:12` and the dataset scripts) and calls `v2e.main()`.

    python -m v2e_amd.launcher /path/to/v2e  -i input.mp4 --dvs346 ...   (or V2E_ROOT=/path/to/v2e)
"""
import importlib
import logging
import os
import sys

logger = logging.getLogger(__name__)


def bind(v2e_root=None):
    """Import `v2e` from `v2e_root` (or sys.path as it is) and bind v2e_amd's classes into it.  Returns the module."""
    import v2e_amd
    if v2e_root:
        v2e_root = os.path.abspath(v2e_root)
        if v2e_root not in sys.path:
            sys.path.insert(0, v2e_root)
    v2e = importlib.import_module("v2e")
    emu_mod = importlib.import_module("v2ecore.emulator")
    slomo_mod = importlib.import_module("v2ecore.slomo")
    for m in (v2e, emu_mod):
        m.EventEmulator = v2e_amd.EventEmulator
    for m in (v2e, slomo_mod):
        m.SuperSloMo = v2e_amd.SuperSloMo
    log_modes()
    return v2e


def log_modes():
    """Say which random-number mode and convolution math the bound classes will run, and what the choice costs (the defaults
    reproduce the reference event for event, which is not the fast configuration)."""
    rng = os.environ.get("V2E_AMD_RNG", "tape").lower()
    if rng == "philox":
        logger.warning("v2e_amd: V2E_AMD_RNG=philox -- DVS noise from counter-based Philox streams on the GPU: same pixel model "
                       "and statistics, NOT the reference's torch MT19937 sequence (events differ from a reference run with the "
                       "same seed); about 12x the frame rate of tape mode (15 900 vs 1 300 frames/s at 346x260)")
    else:
        logger.warning("v2e_amd: rng_mode=tape (default) -- the host replays the reference's torch random draws in the reference's "
                       "order, so a seeded run gives the reference's events bit for bit; this bounds the emulator at about "
                       "1 300 frames/s (346x260).  Set V2E_AMD_RNG=philox for in-kernel Philox noise: about 12x faster, "
                       "statistically equivalent, not sequence-identical")
    logger.warning("v2e_amd: SuperSloMo conv_math=%s (V2E_AMD_CONV_MATH: auto | bf16x3 | fp16x2 | f32; 'bf16x3' is the exact float32 "
                   "split at 0.7x the frame rate of 'auto')", os.environ.get("V2E_AMD_CONV_MATH", "auto"))
    return rng


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    root = os.environ.get("V2E_ROOT")
    if argv and os.path.isfile(os.path.join(argv[0], "v2e.py")):
        root = argv.pop(0)
    v2e = bind(root)
    sys.argv = [os.path.join(root or "", "v2e.py")] + argv
    return v2e.main()


if __name__ == "__main__":
    main()
