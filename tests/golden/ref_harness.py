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


def install_torchvision_stub():
    """`torchvision.transforms` as the reference's slomo.py:148-162 uses it (Compose, ToTensor, Normalize, ToPILImage
    on single-channel uint8 PIL images / float tensors), restated from torchvision's functional code: torchvision itself
    is not in this image.  Test infrastructure only."""
    import numpy as np
    import torch
    from PIL import Image
    if "torchvision.transforms" in sys.modules and not getattr(sys.modules["torchvision.transforms"], "_v2e_stub", False):
        return

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class ToTensor:  # torchvision.transforms.functional.to_tensor for a PIL image of mode L
        def __call__(self, pic):
            a = np.array(pic, np.uint8, copy=True)
            if a.ndim == 2:
                a = a[:, :, None]
            t = torch.from_numpy(a).permute(2, 0, 1).contiguous()
            return t.to(dtype=torch.float32).div(255)

    class Normalize:  # F.normalize: sub_(mean).div_(std) on a clone, mean/std as tensors of the input dtype
        def __init__(self, mean, std):
            self.mean, self.std = mean, std

        def __call__(self, t):
            t = t.clone()
            mean = torch.as_tensor(self.mean, dtype=t.dtype, device=t.device)
            std = torch.as_tensor(self.std, dtype=t.dtype, device=t.device)
            if mean.ndim == 1:
                mean = mean.view(-1, 1, 1)
            if std.ndim == 1:
                std = std.view(-1, 1, 1)
            return t.sub_(mean).div_(std)

    class ToPILImage:  # F.to_pil_image for a float tensor: pic.mul(255).byte(), [C,H,W] -> HWC, mode L for one channel
        def __call__(self, pic):
            if pic.is_floating_point():
                pic = pic.mul(255).byte()
            a = np.transpose(pic.cpu().numpy(), (1, 2, 0))
            if a.shape[2] == 1:
                return Image.fromarray(a[:, :, 0], mode="L")
            return Image.fromarray(a)

    tv = _mod("torchvision")
    tr = _mod("torchvision.transforms", Compose=Compose, ToTensor=ToTensor, Normalize=Normalize, ToPILImage=ToPILImage,
              _v2e_stub=True)
    tv.transforms = tr


def ref_slomo_cls():
    install_stubs()
    install_torchvision_stub()
    from v2ecore.slomo import SuperSloMo
    return SuperSloMo
