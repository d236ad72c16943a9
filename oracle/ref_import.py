"""ORACLE helper: import the *reference itself* (pure Python) from /root/reference with the
shims SURVEY.md section 8(c) lists.  Works only in the build container (the GPU box has no
/root/reference); used by tests/golden/make_golden.py and by the ``not gpu`` cross-checks
that skip when the tree is absent.  Nothing is written into the reference tree.
"""
import importlib
import os
import sys
import types

REF = os.environ.get("HUPR_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF, "models"))


class _NpProxy:
    """numpy stand-in that maps the removed alias dtype='complex_' -> complex128."""

    def __init__(self, np):
        self._np = np

    def __getattr__(self, name):
        return getattr(self._np, name)

    def zeros(self, shape, dtype=float, **kw):
        if isinstance(dtype, str) and dtype == "complex_":
            dtype = self._np.complex128
        return self._np.zeros(shape, dtype=dtype, **kw)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _install_stubs():
    import numpy as np
    import torch

    sys.dont_write_bytecode = True
    if "cv2" not in sys.modules:
        _stub("cv2")
    if "torchvision" not in sys.modules:
        class Compose:
            def __init__(self, ts):
                self.ts = ts

            def __call__(self, x):
                for t in self.ts:
                    x = t(x)
                return x

        class ToTensor:
            def __call__(self, pic):          # float ndarray HWC -> CHW tensor, no scaling
                a = np.asarray(pic)
                if a.ndim == 2:
                    a = a[:, :, None]
                return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))

        tv = _stub("torchvision")
        tv.transforms = _stub("torchvision.transforms", Compose=Compose, ToTensor=ToTensor)
        tv.utils = _stub("torchvision.utils", save_image=None, make_grid=None)
    if "pycocotools" not in sys.modules:
        pc = _stub("pycocotools")
        pc.coco = _stub("pycocotools.coco", COCO=object)
        pc.cocoeval = _stub("pycocotools.cocoeval", COCOeval=object)
    if "matplotlib" in sys.modules or True:
        try:
            import matplotlib
            matplotlib.use("Agg")
        except Exception:
            pass


def _with_ref_path(fn):
    """Import with /root/reference first on sys.path, then remove it and the reference's
    generic top-level names from sys.modules so they cannot shadow the product's."""
    saved = {k: sys.modules.get(k) for k in ("models", "misc", "datasets", "tools")}
    sys.path.insert(0, REF)
    try:
        return fn()
    finally:
        sys.path.remove(REF)
        for k in list(sys.modules):
            root = k.split(".")[0]
            if root in ("models", "misc", "datasets", "tools"):
                mod = sys.modules[k]
                f = getattr(mod, "__file__", "") or ""
                if f.startswith(REF):
                    del sys.modules[k]
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v


def radar_object():
    """Reference RadarObject instance (preprocessing/process_iwr1843.py:8)."""
    import numpy as np
    _install_stubs()
    pre = os.path.join(REF, "preprocessing")
    sys.path.insert(0, pre)
    try:
        if "plot_utils" not in sys.modules:
            _stub("plot_utils", PlotMaps=None, PlotHeatmaps=None)
        spec = importlib.util.spec_from_file_location("_ref_process_iwr1843",
                                                      os.path.join(pre, "process_iwr1843.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.np = _NpProxy(np)
        return mod.RadarObject()
    finally:
        sys.path.remove(pre)
        sys.modules.pop("plot_utils", None)


def model_module():
    """Returns the reference ``models`` package with .cuda() made a no-op (models/layers.py:112)."""
    import torch
    _install_stubs()
    torch.Tensor.cuda = lambda self, *a, **k: self

    def go():
        return importlib.import_module("models")
    return _with_ref_path(go)


def misc_parts():
    """(LossComputer, generateTarget, get_max_preds, Normalize) from the reference."""
    _install_stubs()

    def go():
        losses = importlib.import_module("misc.losses")
        utils = importlib.import_module("misc.utils")
        metrics = importlib.import_module("misc.metrics")
        base = importlib.import_module("datasets.base")
        return losses.LossComputer, utils.generateTarget, metrics.get_max_preds, base.Normalize
    return _with_ref_path(go)


def refcoco():
    """The reference's pycocotools fork (misc/coco.py + misc/cocoeval.py) as a synthetic package with an empty ``mask``
    submodule (they do ``from . import mask``, coco.py:55, cocoeval.py:7); np.float shim for cocoeval.py:381-382."""
    import importlib.util
    import numpy as np
    np.float = float
    _install_stubs()
    if "_refcoco.coco" in sys.modules:
        return sys.modules["_refcoco.coco"], sys.modules["_refcoco.cocoeval"]
    pkg = types.ModuleType("_refcoco")
    pkg.__path__ = [os.path.join(REF, "misc")]
    sys.modules["_refcoco"] = pkg
    sys.modules["_refcoco.mask"] = types.ModuleType("_refcoco.mask")
    mods = {}
    for name in ("coco", "cocoeval"):
        spec = importlib.util.spec_from_file_location("_refcoco." + name, os.path.join(REF, "misc", name + ".py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules["_refcoco." + name] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return mods["coco"], mods["cocoeval"]


def dataset_module():
    """The reference's ``datasets.dataset`` (HuPR3D_horivert, getDataset) with ``pycocotools`` served by its own fork."""
    coco, cocoeval = refcoco()
    sys.modules["pycocotools.coco"].COCO = coco.COCO
    sys.modules["pycocotools.cocoeval"].COCOeval = cocoeval.COCOeval

    def go():
        return importlib.import_module("datasets.dataset")
    return _with_ref_path(go)


def load_cfg():
    """Reference config/mscsa_prgcn.yaml -> attribute tree, built as main.py:7-13 does."""
    import yaml

    class Obj:
        def __init__(self, d):
            for a, b in d.items():
                if isinstance(b, (list, tuple)):
                    setattr(self, a, [Obj(x) if isinstance(x, dict) else x for x in b])
                else:
                    setattr(self, a, Obj(b) if isinstance(b, dict) else b)

    with open(os.path.join(REF, "config", "mscsa_prgcn.yaml")) as f:
        return Obj(yaml.safe_load(f))
