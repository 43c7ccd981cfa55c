"""Import shim that lets the *unmodified* reference modules under ``/root/reference`` import in
the build container, where mmcv / detectron2 / numba / ... are not installed.

ORACLE TOOLING: used only by ``oracle/make_golden.py`` (and ``tests/test_reference_live.py``
when ``/root/reference`` is present).  Never imported by the product and never needed on the
GPU box - the goldens it helps to produce are committed under ``tests/golden/``.

The reference's hot-path files import those packages at module top
(``core/catre/models/CATRE_disR_shared.py:8-12``, ``heads/*.py:4-7``,
``lib/torch_utils/layers/layer_utils.py:8-9``) but use only a handful of symbols, which get
real semantics below; everything else resolves to an inert auto-attribute module.
"""
import importlib.abc
import importlib.machinery
import importlib.util
import os
import sys
import types
from unittest import mock

REFERENCE_ROOT = os.environ.get("CATRE_REFERENCE_ROOT", "/root/reference")

_STUB_ROOTS = {
    "mmcv", "detectron2", "transforms3d", "numba", "fvcore", "timm", "pytorch_lightning", "fairscale",
    "IPython", "cv2", "loguru", "setproctitle", "tensorboardX", "pycocotools", "imgaug", "open3d",
    "horovod", "termcolor", "yacs", "pytorch3d", "plyfile", "png", "imageio", "pyassimp", "OpenGL",
    "glumpy", "vispy", "pyrender", "ruamel", "ujson", "thop", "chardet", "skimage", "omegaconf",
    "apex", "deepspeed", "kornia", "ipdb", "h5py", "albumentations", "imagecorruptions", "matplotlib",
    "PIL", "tabulate", "tqdm", "mmdet", "pytz", "torchvision", "seaborn", "trimesh", "pyglet",
    "progressbar", "absl", "einops", "lmdb",
}


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        m = mock.MagicMock(name=f"{self.__name__}.{name}")
        setattr(self, name, m)
        return m


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        root = fullname.split(".")[0]
        if root in _STUB_ROOTS and root not in _REAL:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Stub(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


_REAL = set()
_installed = False


def install():
    """Idempotent.  After this, ``import core.catre.models...`` works."""
    global _installed
    if _installed:
        return
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    # keep genuinely importable packages real
    for root in list(_STUB_ROOTS):
        try:
            if importlib.util.find_spec(root) is not None:
                _REAL.add(root)
        except (ImportError, ValueError):
            pass
    sys.meta_path.insert(0, _Finder())

    import torch
    import torch.nn as nn

    import mmcv  # stub
    import mmcv.cnn  # stub

    def normal_init(module, mean=0, std=1, bias=0):
        if getattr(module, "weight", None) is not None:
            nn.init.normal_(module.weight, mean, std)
        if getattr(module, "bias", None) is not None:
            nn.init.constant_(module.bias, bias)

    def constant_init(module, val, bias=0):
        if getattr(module, "weight", None) is not None:
            nn.init.constant_(module.weight, val)
        if getattr(module, "bias", None) is not None:
            nn.init.constant_(module.bias, bias)

    mmcv.cnn.normal_init = normal_init
    mmcv.cnn.constant_init = constant_init
    mmcv.is_seq_of = lambda seq, t, seq_type=None: isinstance(seq, seq_type or (list, tuple)) and all(
        isinstance(i, t) for i in seq
    )

    import detectron2.layers
    import detectron2.utils.env

    detectron2.layers.cat = lambda tensors, dim=0: torch.cat(tensors, dim)
    detectron2.utils.env.TORCH_VERSION = tuple(int(v) for v in torch.__version__.split(".")[:2])

    import numba

    def _identity_decorator(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    numba.jit = numba.njit = _identity_decorator

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _installed = True


def load_reference_cfg(leaf=None):
    """Emulate ``mmcv.Config.fromfile`` for the shipped experiment config: exec the python
    config, resolve ``_base_`` recursively, merge child over base, honour ``_delete_``.
    Returns a ``catre_amd.config.CfgNode`` (attribute access like mmcv's ConfigDict)."""
    from catre_amd.config import CfgNode

    leaf = leaf or os.path.join(
        REFERENCE_ROOT,
        "configs/catre/NOCS_REAL/aug05_kpsMS_r9d_catreDisR_shared_tspcl_convPerRot_scaleexp_120e.py",
    )

    def load(path):
        scope = {}
        with open(path) as f:
            exec(compile(f.read(), path, "exec"), scope)
        cur = {k: v for k, v in scope.items() if not k.startswith("__") and not isinstance(v, types.ModuleType)}
        bases = cur.pop("_base_", [])
        bases = [bases] if isinstance(bases, str) else bases
        merged = CfgNode()
        for b in bases:
            merged.merge(load(os.path.normpath(os.path.join(os.path.dirname(path), b))))
        merged.merge(cur)
        return merged

    cfg = load(leaf)
    # what main_catre.py:63-103 derives before the model factory is called
    opt = cfg.SOLVER.get("OPTIMIZER_CFG", None)
    if opt and "lr" in opt:
        cfg.SOLVER.BASE_LR = opt["lr"]
    return cfg


def build_reference_model(cfg):
    """The reference factory minus the optimizer (``CATRE_disR_shared.py:291-322``)."""
    install()
    import copy

    from core.catre.models.CATRE_disR_shared import CATRE_disR_shared
    from core.catre.models.model_utils import get_rot_head, get_ts_head
    from core.catre.models.net_factory import PCLNETS

    init = copy.deepcopy(cfg.MODEL.CATRE.PCLNET.INIT_CFG)
    pcl_net = PCLNETS[init.pop("type")](**init)
    rot_head, _ = get_rot_head(cfg)
    ts_head, _ = get_ts_head(cfg)
    return CATRE_disR_shared(cfg, pcl_net, rot_head, ts_head)
