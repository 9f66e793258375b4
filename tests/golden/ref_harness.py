"""Import the *reference* v2e (read-only at /root/reference) in-process.

Test/fixture-generation infrastructure only.  The reference's
hot path needs only torch+numpy; its other imports (cv2, h5py, numba, ...) are
absent in this image and never called on the hot path, so they are stubbed.
Nothing in the product package imports this module, and nothing that runs on the
GPU box does either (/root/reference does not exist there).
"""
import os
import sys
import types

REF_ROOT = os.environ.get("V2E_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "v2ecore"))


class _Any:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Any()

    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return _Any()


def _mod(name, **kw):
    m = types.ModuleType(name)
    m.__dict__.update(kw)
    sys.modules[name] = m
    return m


def _ga(n):
    if n.startswith("__"):
        raise AttributeError(n)
    return _Any()


_installed = False


def install_stubs():
    global _installed
    if _installed:
        return
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    for name in ("cv2", "dv_processing", "easygui"):
        if name not in sys.modules:
            _mod(name, __getattr__=_ga)
    if "h5py" not in sys.modules:
        _mod("h5py", File=_Any)
    if "screeninfo" not in sys.modules:
        _mod("screeninfo", get_monitors=lambda: [])
    if "engineering_notation" not in sys.modules:
        _mod("engineering_notation", EngNumber=lambda x: x)
    nj = lambda *a, **k: a[0] if (len(a) == 1 and callable(a[0]) and not k) else (lambda f: f)
    if "numba" not in sys.modules:
        _mod("numba", njit=nj, jit=nj)
    try:
        import tkinter  # noqa: F401
    except Exception:
        _mod("tkinter", __getattr__=_ga)
        _mod("tkinter.filedialog", __getattr__=_ga)
    _installed = True


def ref_emulator_cls():
    install_stubs()
    from v2ecore.emulator import EventEmulator
    return EventEmulator


def ref_model():
    install_stubs()
    import v2ecore.model as model
    return model


def ref_moving_dot():
    install_stubs()
    if "skimage" not in sys.modules:
        _mod("skimage", __getattr__=_ga)
    import importlib
    return importlib.import_module("scripts.moving_dot")
