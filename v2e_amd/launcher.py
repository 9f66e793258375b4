"""Launcher that runs the reference's own `v2e.py` with the MI355X hot path bound in.

`v2e.py:38-39` binds `SuperSloMo` and `EventEmulator` as module globals at import time; nothing in the reference tree
is edited: this module imports `v2e` from a checkout, rebinds the two names (there and in the modules they came from, for
`v2ecore/renderer.py:12` and the dataset scripts) and calls `v2e.main()`.

    python -m v2e_amd.launcher /path/to/v2e  -i input.mp4 --dvs346 ...   (or V2E_ROOT=/path/to/v2e)
"""
import importlib
import os
import sys


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
    return v2e


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
