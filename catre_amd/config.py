"""Config surface for the CATRE hot path.

The reference builds its model from an mmcv ``Config`` (attribute-style nested
dict, reference ``core/catre/main_catre.py:46-48``).  mmcv is not a dependency
here: :class:`CfgNode` gives the same attribute/``get``/item access on plain
nested dicts, and :func:`default_cfg` returns the *resolved* hot-path subset of
``configs/catre/NOCS_REAL/aug05_kpsMS_r9d_catreDisR_shared_tspcl_convPerRot_scaleexp_120e.py:1-135``
merged over ``configs/_base_/catre_base.py:94-232`` (only the keys the path reads).

A real mmcv ``Config`` object works unchanged everywhere a ``CfgNode`` is accepted:
the model code only uses attribute access, ``.get`` and ``in``.
"""
import copy


class CfgNode(dict):
    """dict with attribute access, recursively applied (mmcv ``ConfigDict`` look-alike)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = self._wrap(v)

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, CfgNode):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = self._wrap(v)

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __deepcopy__(self, memo):
        return CfgNode({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def to_dict(self):
        out = {}
        for k, v in self.items():
            out[k] = v.to_dict() if isinstance(v, CfgNode) else v
        return out

    def merge(self, other):
        """Recursive dict merge, child over base; honours mmcv's ``_delete_`` key."""
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict) and not v.get("_delete_", False):
                self[k].merge(v)
            else:
                if isinstance(v, dict):
                    v = {kk: vv for kk, vv in v.items() if kk != "_delete_"}
                self[k] = v
        return self


def default_cfg(num_pcl=1024, num_kps=1024, n_iter=4, device="cuda"):
    """Resolved hot-path config of the shipped NOCS_REAL experiment."""
    return CfgNode(
        INPUT=dict(
            NUM_PCL=num_pcl,
            NUM_KPS=num_kps,
            KPS_TYPE="mean_shape",
            WITH_NEG_AXIS=False,
            ZERO_CENTER_INPUT=True,
            # train-side batch glue (configs/_base_/catre_base.py:44-86 under the NOCS_REAL experiment file :10-33)
            INIT_POSE_TYPE_TRAIN=["gt_noise"],
            INIT_SCALE_TYPE_TRAIN=["gt_noise"],
            NOISE_ROT_STD_TRAIN=(10, 5, 2.5, 1.25),
            NOISE_ROT_MAX_TRAIN=45,
            NOISE_TRANS_STD_TRAIN=[(0.02, 0.02, 0.02), (0.01, 0.01, 0.01), (0.005, 0.005, 0.005)],
            INIT_TRANS_MIN_Z=0.1,
            NOISE_SCALE_STD_TRAIN=[(0.01, 0.01, 0.01), (0.005, 0.005, 0.005), (0.002, 0.002, 0.002)],
            INIT_SCALE_MIN=0.04,
            RANDOM_TRANS_MIN=[-0.35, -0.35, 0.5],
            RANDOM_TRANS_MAX=[0.35, 0.35, 1.3],
            RANDOM_SCALE_MIN=[0.04, 0.04, 0.04],
            RANDOM_SCALE_MAX=[0.5, 0.3, 0.4],
            CANONICAL_ROT=[(1, 0, 0, 0.5), (0, 0, 1, -0.7)],
            CANONICAL_TRANS=[0, 0, 1.0],
            CANONICAL_SIZE=[0.2, 0.2, 0.2],
            BBOX3D_AUG_PROB=0.5,
            RT_AUG_PROB=0.5,
        ),
        SOLVER=dict(
            IMS_PER_BATCH=16,
            BASE_LR=1e-4,
            OPTIMIZER_CFG=dict(type="Ranger", lr=1e-4, weight_decay=0),
            WEIGHT_DECAY=0.0,
        ),
        MODEL=dict(
            DEVICE=device,
            WEIGHTS="",
            REFINE_SCLAE=True,
            CATRE=dict(
                NAME="CATRE_disR_shared",
                TASK="refine",
                NUM_CLASSES=6,
                N_ITER_TRAIN=n_iter,
                N_ITER_TRAIN_WARM_EPOCH=4,
                N_ITER_TEST=n_iter,
                USE_MTL=False,
                PCLNET=dict(
                    FREEZE=False,
                    INIT_CFG=dict(
                        type="point_net",
                        num_points=num_pcl,
                        global_feat=False,
                        feature_transform=True,
                        out_dim=1024,
                    ),
                ),
                ROT_HEAD=dict(
                    FREEZE=False,
                    ROT_TYPE="ego_rot6d",
                    CLASS_AWARE=False,
                    INIT_CFG=dict(
                        type="ConvOutPerRotHead",
                        in_dim=1088,
                        num_layers=2,
                        kernel_size=1,
                        feat_dim=256,
                        norm="GN",
                        num_gn_groups=32,
                        act="gelu",
                        num_points=num_pcl + num_kps,
                        rot_dim=3,
                        norm_input=False,
                    ),
                    LR_MULT=1.0,
                    DELTA_T_SPACE="image",
                    DELTA_T_WEIGHT=1.0,
                    T_TRANSFORM_K_AWARE=True,
                    DELTA_Z_STYLE="cosypose",
                    SCLAE_TYPE="iter_add",
                ),
                TS_HEAD=dict(
                    WITH_KPS_FEATURE=False,
                    WITH_INIT_SCALE=True,
                    WITH_INIT_TRANS=False,
                    FREEZE=False,
                    INIT_CFG=dict(
                        type="FC_TransSizeHead",
                        in_dim=1088 + 3,
                        num_layers=2,
                        feat_dim=256,
                        norm="GN",
                        num_gn_groups=32,
                        act="gelu",
                        norm_input=False,
                    ),
                    LR_MULT=1.0,
                ),
                LOSS_CFG=dict(
                    PM_LOSS_TYPE="L1",
                    PM_SMOOTH_L1_BETA=1.0,
                    PM_LOSS_SYM=True,
                    PM_NORM_BY_EXTENT=False,
                    PM_R_ONLY=True,
                    PM_WITH_SCALE=True,
                    PM_DISENTANGLE_T=False,
                    PM_DISENTANGLE_Z=False,
                    PM_T_USE_POINTS=True,
                    PM_USE_BBOX=False,
                    PM_LW=1.0,
                    ROT_LOSS_TYPE="angular",
                    ROT_YAXIS_LOSS_TYPE="L1",
                    ROT_LW=1.0,
                    TRANS_LOSS_TYPE="L1",
                    TRANS_LOSS_DISENTANGLE=True,
                    TRANS_LW=1.0,
                    SCALE_LOSS_TYPE="L1",
                    SCALE_LW=1.0,
                ),
            ),
        ),
    )
