"""Head builders - same names/returns as reference ``core/catre/models/model_utils.py:66-89,144-167``."""
import copy

from .net_factory import HEADS, PCLNETS  # noqa: F401


def get_rot_dim(rot_type):
    """reference model_utils.py:11-25."""
    if rot_type in ["allo_quat", "ego_quat"]:
        return 4
    if rot_type in ["allo_log_quat", "ego_log_quat", "allo_lie_vec", "ego_lie_vec"]:
        return 3
    if rot_type in ["allo_rot6d", "ego_rot6d"]:
        return 6
    raise ValueError(f"Unknown rot_type: {rot_type}")


def _build_head(cfg, head_cfg, num_classes):
    init_cfg = copy.deepcopy(head_cfg.INIT_CFG)
    init_cfg = dict(init_cfg)
    head_type = init_cfg.pop("type")
    init_cfg.update(num_classes=num_classes)
    head = HEADS[head_type](**init_cfg)
    params_lr_list = []
    if head_cfg.get("FREEZE", False):
        for p in head.parameters():
            p.requires_grad = False
    else:
        params_lr_list.append(
            {
                "params": filter(lambda p: p.requires_grad, head.parameters()),
                "lr": float(cfg.SOLVER.BASE_LR) * head_cfg.get("LR_MULT", 1.0),
            }
        )
    return head, params_lr_list


def get_rot_head(cfg):
    net_cfg = cfg.MODEL.CATRE
    rot_num_classes = net_cfg.NUM_CLASSES if net_cfg.ROT_HEAD.get("CLASS_AWARE", False) else 1
    return _build_head(cfg, net_cfg.ROT_HEAD, rot_num_classes)


def get_ts_head(cfg):
    net_cfg = cfg.MODEL.CATRE
    num_classes = net_cfg.NUM_CLASSES if net_cfg.ROT_HEAD.get("CLASS_AWARE", False) else 1
    return _build_head(cfg, net_cfg.TS_HEAD, num_classes)
