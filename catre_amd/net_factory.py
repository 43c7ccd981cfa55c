"""Name -> class registries for the point-cloud encoder and the residual heads.

The config surface selects modules by string (``PCLNET.INIT_CFG.type``, ``ROT_HEAD.INIT_CFG.type``,
``TS_HEAD.INIT_CFG.type``); the registry NAMES and KEYS below are the reference's
(``core/catre/models/net_factory.py:6-13``) so its configs resolve unchanged, the classes are the HIP-backed
mirrors of this package.
"""
from . import heads as _heads
from . import pointnet as _pointnet


def _registry(**entries):
    table = dict(entries)
    for key, cls in table.items():
        if not isinstance(cls, type):
            raise TypeError(f"registry entry {key!r} is not a class")
    return table


# encoder(s) for observed points and the transformed shape prior (shared weights)
PCLNETS = _registry(point_net=_pointnet.PointNetfeat)

# residual heads: translation+size (FC) and per-axis rotation (point-wise conv)
HEADS = _registry(
    FC_TransSizeHead=_heads.FC_TransSizeHead,
    ConvOutPerRotHead=_heads.ConvOutPerRotHead,
)
