"""CPU-side checks: drop-in surface (names, state_dict, factory), config handling, the C-ABI library loads and
exports every symbol the header declares, and the product refuses to run without a HIP device."""
import copy
import ctypes
import os
import re

import pytest
import torch

from catre_amd import hip
from catre_amd.config import CfgNode, default_cfg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "catre_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(catre_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(hip.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(hip.LIB_PATH)
    declared = _header_functions()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/catre_hip.h but not exported"
    assert set(declared) == set(hip.EXPORTED_SYMBOLS), set(declared) ^ set(hip.EXPORTED_SYMBOLS)
    lib.catre_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.catre_version()


def test_header_is_plain_c_and_a_c_program_links_against_the_library(tmp_path):
    """The boundary is a C ABI: include/catre_hip.h compiles as C99 (no C++, no torch types) and a C program that only
    includes it links against libcatre_hip.so and can call the entry points that need no GPU."""
    import shutil
    import subprocess

    from catre_amd import hip

    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = hip.LIB_PATH
    src = tmp_path / "abi.c"
    src.write_text(
        '#include <stdio.h>\n#include <string.h>\n#include "catre_hip.h"\n'
        "int main(void) {\n"
        "  catre_opts o; memset(&o, 0, sizeof o);\n"
        "  if (sizeof(catre_opts) != 72) return 2;\n"
        "  if (catre_workspace_bytes(2, 1024, 1024) == 0) return 3;\n"
        "  if (catre_refine_k(NULL, NULL, NULL, NULL, NULL, NULL, &o, NULL, NULL, NULL, 0, 2, 1024, 1024, 4, NULL) != CATRE_ERR_BAD_ARG) return 4;\n"
        '  printf("%s %s\\n", catre_version(), catre_status_string(CATRE_ERR_WORKSPACE));\n'
        "  return 0;\n}\n")
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", f"-I{root}/include", str(src), "-o", str(exe),
                    f"-L{os.path.dirname(lib)}", "-lcatre_hip", f"-Wl,-rpath,{os.path.dirname(lib)}"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    assert "catre_hip" in out and "workspace" in out.lower()


def test_size_queries_without_gpu():
    lib = hip.load()
    assert lib.catre_workspace_bytes(0, 1024, 1024) == 0
    assert lib.catre_workspace_bytes(256, 1024, 1024) > 1 << 30
    small, big = lib.catre_workspace_bytes(1, 64, 64), lib.catre_workspace_bytes(2, 64, 64)
    assert 0 < small < big
    assert lib.catre_packed_floats(1024, 1024, 1091) > 1024 * 512
    assert lib.catre_status_string(-2).decode().startswith("workspace")
    assert ctypes.sizeof(hip.CatreOpts) == 72 and ctypes.sizeof(hip.CatrePoints) == 88


def test_param_enum_matches_header():
    src = open(os.path.join(ROOT, "include", "catre_hip.h")).read()
    body = src[src.index("typedef enum catre_param {"):src.index("} catre_param;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = re.findall(r"CATRE_P_[A-Z0-9_]+", body)
    assert names[-1] == "CATRE_P_COUNT" and len(names) - 1 == hip.CATRE_P_COUNT == len(hip.PARAM_KEYS)


def test_state_dict_surface_matches_reference_listing():
    """Keys / shapes of SURVEY.md section 8b (probe listing of the reference model)."""
    from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes

    cfg = default_cfg(device="cpu")
    model, opt = build_model_optimizer(cfg, is_test=False)
    sd = model.state_dict()
    exp = expected_state_shapes(cfg)
    assert list(sd) and set(sd) == set(exp)
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(exp[k]), k
    assert len(sd) == 74 and sum(v.numel() for v in sd.values()) == 4298711
    # every tensor the kernels read is a state_dict entry; the 6 never-used `norm` tensors are not read
    assert set(hip.PARAM_KEYS) <= set(sd)
    assert sorted(set(sd) - set(hip.PARAM_KEYS)) == sorted(
        f"{p}.norm.{w}" for p in ("rot_head.rot_head_x", "rot_head.rot_head_y", "ts_head") for w in ("weight", "bias"))
    # param groups: pcl_net @ BASE_LR, rot_head / ts_head @ BASE_LR * LR_MULT  (reference :306-315)
    assert [len(g["params"]) for g in opt.param_groups] == [32, 28, 14]
    assert all(abs(g["lr"] - 1e-4) < 1e-12 for g in opt.param_groups)
    # init recipe of the heads (conv_out_per_rot_head.py:117-124, fc_trans_size_head.py:50-59)
    assert float(sd["rot_head.rot_head_x.layers.0.weight"].std()) < 2e-3
    assert float(sd["rot_head.rot_head_x.layers.0.bias"].abs().max()) == 0.0
    assert 5e-3 < float(sd["ts_head.fc_t.weight"].std()) < 2e-2
    assert torch.equal(sd["ts_head.linears.1.weight"], torch.ones(256))
    _, none_opt = build_model_optimizer(cfg, is_test=True)
    assert none_opt is None


def test_registries_and_names():
    import catre_amd.CATRE_disR_shared as mod
    from catre_amd.net_factory import HEADS, PCLNETS

    assert set(PCLNETS) == {"point_net"} and set(HEADS) == {"FC_TransSizeHead", "ConvOutPerRotHead"}
    assert hasattr(mod, "build_model_optimizer") and hasattr(mod, "CATRE_disR_shared")
    cfg = default_cfg(device="cpu")
    cfg.MODEL.CATRE.NAME = "something_else"
    with pytest.raises(AssertionError):
        mod.build_model_optimizer(cfg, is_test=True)


def test_opts_from_cfg_flag_mapping():
    from catre_amd.runtime import opts_from_cfg

    cfg = default_cfg(device="cpu")
    o = opts_from_cfg(cfg)
    assert (o.delta_t_space_3d, o.delta_z_deepim, o.k_aware, o.scale_mul, o.scale_base_mean, o.is_allo) == (0, 0, 1, 0, 0, 0)
    assert (o.with_kps_feature, o.with_init_scale, o.with_init_trans, o.ts_in_dim) == (0, 1, 0, 1091)
    assert o.zero_center == 1 and o.refine_scale == 1 and abs(o.delta_t_weight - 1.0) < 1e-12
    cfg.MODEL.CATRE.ROT_HEAD.SCLAE_TYPE = "mean_mul"
    cfg.MODEL.CATRE.ROT_HEAD.ROT_TYPE = "allo_rot6d"
    cfg.MODEL.CATRE.ROT_HEAD.DELTA_T_SPACE = "3D"
    cfg.MODEL.CATRE.TS_HEAD.WITH_KPS_FEATURE = True
    o = opts_from_cfg(cfg)
    assert (o.scale_mul, o.scale_base_mean, o.is_allo, o.delta_t_space_3d, o.ts_in_dim) == (1, 1, 1, 1, 2179)
    cfg.MODEL.CATRE.ROT_HEAD.DELTA_T_SPACE = "nope"
    with pytest.raises(ValueError):
        opts_from_cfg(cfg)
    cfg = default_cfg(device="cpu")
    for rt, want in (("ego_quat", hip.ROT_QUAT), ("allo_log_quat", hip.ROT_LOG_QUAT), ("ego_lie_vec", hip.ROT_LIE_VEC),
                     ("allo_rot6d", hip.ROT_6D)):
        cfg.MODEL.CATRE.ROT_HEAD.ROT_TYPE = rt
        o = opts_from_cfg(cfg)
        assert o.rot_type == want and o.is_allo == int(rt.startswith("allo"))
    cfg.MODEL.CATRE.ROT_HEAD.ROT_TYPE = "ego_euler"
    with pytest.raises(ValueError, match="Unknown rot_type"):  # model_utils.py:24
        opts_from_cfg(cfg)
    # in_dim inconsistent with the gathered features is caught at construction
    cfg = default_cfg(device="cpu")
    cfg.MODEL.CATRE.TS_HEAD.WITH_INIT_SCALE = False
    from catre_amd.CATRE_disR_shared import build_model_optimizer
    with pytest.raises(ValueError):
        build_model_optimizer(cfg, is_test=True)


def test_no_cpu_fallback():
    """The product path must fail loudly without a HIP device - never route to the oracle or to torch ops."""
    from catre_amd.CATRE_disR_shared import build_model_optimizer

    cfg = default_cfg(num_pcl=8, num_kps=8, device="cpu")
    model, _ = build_model_optimizer(cfg, is_test=True)
    x = torch.zeros(1, 3, 8)
    with torch.no_grad():
        with pytest.raises(hip.CatreHipError):
            model(x, x, torch.zeros(1, 3, 4), torch.zeros(1, 3))
        with pytest.raises(hip.CatreHipError):
            model.pcl_net(x)
    with pytest.raises(hip.CatreHipError):  # grad-enabled (training) call: HIP ops only, no torch-op fallback
        model(x, x, torch.zeros(1, 3, 4), torch.zeros(1, 3))
    src = "".join(open(os.path.join(ROOT, "catre_amd", f)).read() for f in os.listdir(os.path.join(ROOT, "catre_amd"))
                  if f.endswith(".py"))
    assert "import oracle" not in src and "from oracle" not in src, "product code must not import the oracle"


def test_model_copies_do_not_share_runtime():
    from catre_amd.CATRE_disR_shared import build_model_optimizer

    model, _ = build_model_optimizer(default_cfg(device="cpu"), is_test=True)
    model._rt = object()
    clone = copy.deepcopy(model)
    assert clone._rt is None and clone._opts.ts_in_dim == 1091
    assert clone.pcl_net.conv4.weight.data_ptr() != model.pcl_net.conv4.weight.data_ptr()


def test_cfgnode_merge_semantics():
    base = CfgNode(A=dict(x=1, y=dict(z=2)), B=3)
    base.merge(dict(A=dict(y=dict(w=5)), C=7))
    assert base.A.x == 1 and base.A.y.z == 2 and base.A.y.w == 5 and base.C == 7
    base.merge(dict(A=dict(_delete_=True, q=1)))
    assert dict(base.A) == {"q": 1}
    assert base.get("missing", 9) == 9 and "B" in base


def test_recipe_is_deterministic():
    from catre_amd import synth
    from catre_amd.CATRE_disR_shared import expected_state_shapes

    shapes = expected_state_shapes(default_cfg(device="cpu"))
    a = synth.recipe_state_dict(shapes)
    b = synth.recipe_state_dict(shapes)
    assert all(torch.equal(a[k], b[k]) for k in a)
    c = synth.make_inputs(3, 64, 32, seed=4)
    d = synth.make_inputs(3, 64, 32, seed=4)
    assert all(torch.equal(c[k], d[k]) for k in c)
    assert c["pcl"].shape == (3, 64, 3) and c["obj_kps"].shape == (3, 32, 3) and c["obj_pose_est"].shape == (3, 3, 4)


def test_empty_batch_returns_empty_outputs_without_touching_the_device():
    """The evaluator skips frames without instances (catre_evaluator.py:280-281); a zero-object batch is still a valid
    call and must not need the GPU."""
    import torch

    from catre_amd.CATRE_disR_shared import build_model_optimizer
    from catre_amd.config import default_cfg

    cfg = default_cfg(num_pcl=64, num_kps=32, n_iter=2, device="cpu")
    model, _ = build_model_optimizer(cfg, is_test=True)
    model.eval()
    z = torch.zeros
    with torch.no_grad():
        out = model(z(0, 3, 64), z(0, 3, 32), init_pose=z(0, 3, 4), init_scale=z(0, 3), K_zoom=z(0, 3, 3), cur_iter=1)
    assert out["pose_1"].shape == (0, 3, 4) and out["scale_1"].shape == (0, 3)
    out = model.refine({"pcl": z(0, 64, 3), "obj_kps": z(0, 32, 3), "obj_pose_est": z(0, 3, 4), "obj_scale_est": z(0, 3),
                        "K": z(0, 3, 3)})
    assert sorted(out) == ["pose_0", "pose_1", "pose_2", "scale_0", "scale_1", "scale_2"] and out["pose_2"].shape == (0, 3, 4)


def test_integration_md_stub_struct_matches_the_header():
    """The ctypes stub in INTEGRATION.md section 2 declares catre_opts field for field like include/catre_hip.h (a shorter
    struct would make the library read past it)."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sect = md[md.index("## 2. Binding the C ABI directly"):md.index("## 3. Entry points")]
    code = re.search(r"```python\n(.*?)```", sect, flags=re.S).group(1)
    ns = {}
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        exec(compile(code, "INTEGRATION.md#2", "exec"), ns)
    finally:
        os.chdir(cwd)
    assert [(n, t) for n, t in ns["catre_opts"]._fields_] == [(n, t) for n, t in hip.CatreOpts._fields_]
    src = open(os.path.join(ROOT, "include", "catre_hip.h")).read()
    body = src[src.index("typedef struct catre_opts {"):src.index("} catre_opts;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"(?:int32_t|float)\s+(\w+);", body)
    assert fields == [n for n, _ in hip.CatreOpts._fields_]


def test_shard_batch_slices_only_per_object_entries():
    from catre_amd.sharding import shard_batch

    B = 4
    batch = {"pcl": torch.zeros(B, 8, 3), "obj_cls": torch.arange(B), "sym_info": [None] * B,
             "img": torch.zeros(B, 3, 2, 2),      # per IMAGE: happens to have B entries, must not be sliced
             "depth_obs": torch.zeros(B, 2, 2), "note": "x"}
    s = shard_batch(batch, 1, 2)
    assert s["pcl"].shape[0] == 2 and s["obj_cls"].tolist() == [2, 3] and len(s["sym_info"]) == 2
    assert s["img"].shape[0] == B and s["depth_obs"].shape[0] == B and s["note"] == "x"
    assert shard_batch(batch, 0, 2, extra_keys=("img",))["img"].shape[0] == 2
    with pytest.raises(ValueError, match="per-object entry"):
        shard_batch(dict(batch, obj_cls=torch.arange(3)), 0, 2)
    # batching.py:40: concatenated over instances -> per object
    assert shard_batch(dict(batch, last_frame_poses=torch.zeros(B, 3, 4)), 1, 2)["last_frame_poses"].shape[0] == 2
    # an unlisted entry with exactly B rows is ambiguous: refuse, unless the caller says which it is
    extra = dict(batch, my_feat=torch.zeros(B, 5))
    with pytest.raises(ValueError, match="not in PER_OBJECT_KEYS"):
        shard_batch(extra, 0, 2)
    assert shard_batch(extra, 0, 2, extra_keys=("my_feat",))["my_feat"].shape[0] == 2
    assert shard_batch(extra, 0, 2, per_image_keys=("my_feat",))["my_feat"].shape[0] == B
    assert shard_batch(extra, 0, 1)["my_feat"].shape[0] == B         # world 1: nothing to pair wrongly


def test_optimizer_factory_follows_the_reference_builder():
    """build_optimizer_with_params (core/utils/solver_utils.py:75-87): dict or string OPTIMIZER_CFG, every keyword reaches
    the optimizer, unknown names raise, gradient clipping wraps step()."""
    from catre_amd.CATRE_disR_shared import build_model_optimizer

    cfg = default_cfg(device="cpu")
    cfg.SOLVER.OPTIMIZER_CFG = dict(type="SGD", lr=3e-3, momentum=0.85, nesterov=True, weight_decay=1e-4)
    _, opt = build_model_optimizer(cfg, is_test=False)
    assert isinstance(opt, torch.optim.SGD)
    g = opt.param_groups[0]
    assert (g["momentum"], g["nesterov"], g["weight_decay"]) == (0.85, True, 1e-4)
    assert abs(opt.param_groups[0]["lr"] - 1e-4) < 1e-12  # per-group lr (BASE_LR) overrides the default, as in the reference
    cfg.SOLVER.OPTIMIZER_CFG = "dict(type='AdamW', lr=1e-4, betas=(0.8, 0.9), weight_decay=0.01)"
    _, opt = build_model_optimizer(cfg, is_test=False)
    assert isinstance(opt, torch.optim.AdamW) and opt.param_groups[0]["betas"] == (0.8, 0.9)
    cfg.SOLVER.OPTIMIZER_CFG = ""
    with pytest.raises(RuntimeError, match="OPTIMIZER_CFG"):
        build_model_optimizer(cfg, is_test=False)
    cfg.SOLVER.OPTIMIZER_CFG = dict(type="NoSuchOpt", lr=1.0)
    with pytest.raises(ValueError, match="Unknown optimizer name"):
        build_model_optimizer(cfg, is_test=False)
    # SOLVER.CLIP_GRADIENTS (lib/torch_utils/solver/grad_clip_d2.py): full-model norm clip before the step
    cfg.SOLVER.OPTIMIZER_CFG = dict(type="SGD", lr=1.0)
    cfg.SOLVER.CLIP_GRADIENTS = dict(ENABLED=True, CLIP_TYPE="full_model", CLIP_VALUE=0.5, NORM_TYPE=2.0)
    model, opt = build_model_optimizer(cfg, is_test=False)
    assert type(opt).__name__ == "SGDWithGradientClip" and isinstance(opt, torch.optim.SGD)
    for p in model.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    total = torch.sqrt(sum((p.grad ** 2).sum() for p in model.parameters()))
    assert abs(float(total) - 0.5) < 1e-4


def test_freeze_unused_norm_knob_takes_the_six_dead_tensors_out_of_the_trainable_set():
    """MODEL.CATRE.FREEZE_UNUSED_NORM (opt-in, bench.py `ddp_world1`): the six `norm.*` tensors no forward uses stay in the
    state_dict but leave the trainable set and the optimizer's groups - DDP's find_unused_parameters path then has nothing
    to wait for.  Default: the reference's surface, 74 trainable tensors."""
    from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes

    cfg = default_cfg(device="cpu")
    cfg.SOLVER.OPTIMIZER_CFG = dict(type="SGD", lr=1e-3)
    model, opt = build_model_optimizer(cfg, is_test=False)
    assert sum(p.requires_grad for p in model.parameters()) == 74
    assert sum(len(g["params"]) for g in opt.param_groups) == 74
    cfg.MODEL.CATRE.FREEZE_UNUSED_NORM = True
    model, opt = build_model_optimizer(cfg, is_test=False)
    frozen = sorted(k for k, p in model.named_parameters() if not p.requires_grad)
    assert frozen == sorted(f"{h}.norm.{w}" for h in ("rot_head.rot_head_x", "rot_head.rot_head_y", "ts_head")
                            for w in ("weight", "bias"))
    assert sum(len(g["params"]) for g in opt.param_groups) == 68
    assert set(model.state_dict()) == set(expected_state_shapes(cfg))      # still 74 keys: strict loading keeps working


def test_pretrained_pcl_net_checkpoint_is_loaded(tmp_path):
    from catre_amd.CATRE_disR_shared import build_model_optimizer

    cfg = default_cfg(device="cpu")
    ref, _ = build_model_optimizer(cfg, is_test=True)
    path = str(tmp_path / "pcl.pth")
    torch.save({"state_dict": {"module." + k: v + 1.0 for k, v in ref.pcl_net.state_dict().items()}}, path)
    cfg.MODEL.CATRE.PCLNET.PRETRAINED = path
    model, _ = build_model_optimizer(cfg, is_test=True)
    a, b = ref.pcl_net.state_dict(), model.pcl_net.state_dict()
    assert all(torch.equal(a[k] + 1.0, b[k]) for k in a)


def test_loss_type_names_of_the_reference_are_accepted():
    from catre_amd.losses import _loss_cfg_struct

    cfg = default_cfg(device="cpu")
    lc = cfg.MODEL.CATRE.LOSS_CFG
    lc.TRANS_LOSS_TYPE, lc.SCALE_LOSS_TYPE, lc.ROT_YAXIS_LOSS_TYPE = "L2", "L2", "angular"
    c = _loss_cfg_struct(cfg)
    assert (c.trans_mse, c.scale_mse, c.yaxis_smooth) == (2, 2, 3)
    lc.ROT_YAXIS_LOSS_TYPE = "L2"
    assert _loss_cfg_struct(cfg).yaxis_smooth == 2
    for field, msg in (("TRANS_LOSS_TYPE", "Unknown trans loss type"), ("SCALE_LOSS_TYPE", "Unknown scale loss type"),
                       ("ROT_YAXIS_LOSS_TYPE", "Unknown rot yaxis loss type"), ("ROT_LOSS_TYPE", "Unknown rot loss type")):
        cfg2 = default_cfg(device="cpu")
        cfg2.MODEL.CATRE.LOSS_CFG[field] = "huber"
        with pytest.raises(ValueError, match=msg):
            _loss_cfg_struct(cfg2)


def test_runtime_reads_live_parameters_through_cached_slots():
    """HipRuntime resolves the 74 parameters through cached `module._parameters` slots (no module-tree walk per forward).
    A re-assigned Parameter and a replaced sub-module must still be seen - the packed weights are keyed on what this
    returns."""
    from catre_amd.CATRE_disR_shared import build_model_optimizer

    cfg = default_cfg(device="cpu")
    model, _ = build_model_optimizer(cfg, is_test=True)
    rt = model._runtime()
    want = dict(model.named_parameters())
    got = rt._live_params()
    assert len(got) == len(hip.PARAM_KEYS)
    assert all(t is want.get(k) for k, t in zip(hip.PARAM_KEYS, got))
    i = hip.PARAM_KEYS.index("pcl_net.conv4.weight")
    model.pcl_net.conv4.weight = torch.nn.Parameter(torch.zeros_like(model.pcl_net.conv4.weight))
    assert rt._live_params()[i] is model.pcl_net.conv4.weight
    fresh = copy.deepcopy(model.ts_head)
    model.ts_head = fresh
    j = hip.PARAM_KEYS.index("ts_head.fc_t.weight")
    assert rt._live_params()[j] is fresh.fc_t.weight
    want = dict(model.named_parameters())
    assert all(t is want.get(k) for k, t in zip(hip.PARAM_KEYS, rt._live_params()))


def test_design_roofline_table_is_generated_from_the_committed_profiles():
    """DESIGN.md section 3's per-kernel table is emitted by profiles/make_tables.py from the CSVs under profiles/ - the
    document cannot quote a number the committed rocprofv3 summaries do not hold."""
    import importlib.util
    import re

    spec = importlib.util.spec_from_file_location("make_tables", os.path.join(ROOT, "profiles", "make_tables.py"))
    mt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mt)
    doc = open(os.path.join(ROOT, "DESIGN.md")).read()
    m = re.search(r"<!-- BEGIN GENERATED by profiles/make_tables.py (\w+) -->.*?<!-- END GENERATED -->", doc, flags=re.S)
    assert m, "DESIGN.md lost its GENERATED block"
    assert m.group(0) == mt.render(m.group(1)), "DESIGN.md table is stale: python profiles/make_tables.py <tag> --write"
    for row in ("| `k_trunk4`", "| `k_stn3d_pair`", "| `k_rot_l1<1>`", "| `k_trunk4<true>`", "| `k_rot_l1<1, true>`", "| `k_rot_l1_bwd`", "| `k_trunk_bf2`",
                "| `k_trunk_split<1>`"):
        assert row in m.group(0), f"roofline table lost its {row} row (kernel renamed? see make_tables.canon)"


def test_bench_train_flop_count_comes_from_the_committed_pmc_pass():
    """`train_fp32.path_frac_of_mfma_peak` divides COUNTED MFMA work (newest profiles/rNN_train_pmc_summary.csv: dispatches x
    SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 FLOP per iteration) by measured time - no hand-written FLOP budget."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    flop, src = bench.train_flops_per_iteration()
    assert src and src.startswith("profiles/r") and src.endswith("_train_pmc_summary.csv")
    # forward 1.10 TFLOP (4.315 GFLOP x 256) + rot-head backward 0.34 + row-sparse encoder backward < 0.1
    assert 1.45e12 < flop < 1.65e12, flop


def test_training_kernel_knobs_are_scoped_per_call_not_module_globals():
    """the A/B switches of the training forward are fields of ``cfg.MODEL.CATRE.TRAIN_KERNELS`` applied
    around ONE model's forward (thread-local, restored on exit), not module attributes."""
    import threading

    from catre_amd import train_ops as T

    base = T.knobs()
    assert all(isinstance(v, bool) for v in base)
    with T.train_kernels(dict(fused_lp_rot=False), lp_rot_bf16_rows=False):
        assert T.knobs().fused_lp_rot is False and T.knobs().lp_rot_bf16_rows is False
        seen = []
        t = threading.Thread(target=lambda: seen.append(T.knobs()))   # another thread keeps the defaults
        t.start()
        t.join()
        assert seen[0] == base
        with T.train_kernels(None):
            assert T.knobs().fused_lp_rot is False
    assert T.knobs() == base
    with pytest.raises(ValueError):
        T.train_kernels(dict(no_such_knob=True))
    for name in ("FUSED_LP_ROT", "LP_ROT_BF16_ROWS", "LP_ROT_FUSE_GN0", "SPLIT_L0_SP"):
        assert not hasattr(T, name)


def test_loss_dict_sum_chain_returns_the_precomputed_running_sums():
    """catre_amd.losses._LossTerm on plain CPU tensors (the mechanics need no GPU): `sum(dict.values())` walks the chain
    0 + v0, prefix0 + v1, ... and gets the precomputed tensors back (attached to the same autograd graph); any other
    arithmetic on the values is plain torch and gives plain tensors."""
    import torch

    from catre_amd.losses import _LossTerm

    l = torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0, 6.0], requires_grad=True)
    order = [0, 2, 5]
    pre = torch.cumsum(l[order], 0)
    tok = object()
    sums = [t.as_subclass(_LossTerm) for t in pre.unbind(0)]
    for k, t in enumerate(sums):
        t._chain_tok, t._sum_pos = tok, k
    ld = {}
    for k, (name, i) in enumerate(zip("abc", order)):
        t = l.unbind(0)[i].as_subclass(_LossTerm)
        t._chain_tok, t._term_pos, t._next_sum = tok, k, sums[k]
        ld[name] = t
    tot = sum(ld.values())
    assert tot is sums[2] and float(tot.detach()) == 10.0
    tot.backward()
    assert l.grad.tolist() == [1.0, 0.0, 1.0, 0.0, 0.0, 1.0]
    assert ld["c"] + (ld["b"] + (ld["a"] + 0)) is sums[2]                 # either operand order
    acc = 0
    for v in ld.values():
        acc += v                                                           # the loop form of the same sum
    assert acc is sums[2]
    acc = 0
    acc += ld["a"]
    acc += ld["c"]                                                         # off the chain: a plain out-of-place sum
    assert type(acc) is torch.Tensor and float(acc.detach()) == 7.0 and float(sums[0].detach()) == 1.0
    assert type(ld["a"] + ld["c"]) is torch.Tensor and float((ld["a"] + ld["c"]).detach()) == 7.0   # off the chain
    assert type(ld["b"] * 2) is torch.Tensor and type(torch.stack(list(ld.values()))) is torch.Tensor
    assert type(1 + ld["a"]) is torch.Tensor and type(sums[0] + ld["c"]) is torch.Tensor              # not the next step
    other = object()
    foreign = l.unbind(0)[1].as_subclass(_LossTerm)
    foreign._chain_tok, foreign._term_pos, foreign._next_sum = other, 1, sums[1]
    assert type(sums[0] + foreign) is torch.Tensor                                                    # another dict's term
    assert {k: round(float(v.detach()), 6) for k, v in ld.items()} == {"a": 1.0, "b": 3.0, "c": 6.0}
    # what logging code does with loss values: torch's own `type(t) is Tensor` paths see plain tensors
    assert f"{sums[2].detach():.3f}" == "10.000" and "{:.2f}".format(ld["b"].detach()) == "3.00" and ld["c"].item() == 6.0
    assert bool(torch.isfinite(tot)) and max(ld.values()) is ld["c"]
