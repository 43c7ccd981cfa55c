"""Registries, same names and keys as reference ``core/catre/models/net_factory.py:6-13``."""
from .heads import ConvOutPerRotHead, FC_TransSizeHead
from .pointnet import PointNetfeat

PCLNETS = {
    "point_net": PointNetfeat,
}

HEADS = {
    "FC_TransSizeHead": FC_TransSizeHead,
    "ConvOutPerRotHead": ConvOutPerRotHead,
}
