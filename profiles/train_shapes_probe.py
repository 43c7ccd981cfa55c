"""Three training iterations at assorted (B, N, M) in fp32 and under autocast: finite parameters, falling losses, the
autocast losses next to the fp32 ones (a smoke probe for shapes the parity tests do not visit)."""
import logging, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.disable(logging.CRITICAL)
from catre_amd import synth
from catre_amd.batching import batch_updater_test
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg
from catre_amd.synth import y_axis_symmetries
for (B, N, M) in ((8, 2048, 1024), (3, 64, 192), (33, 128, 64), (1, 1024, 1024), (5, 4096, 64)):
    for amp in (False, True):
        cfg = default_cfg(num_pcl=N, num_kps=M, device="cuda:0")
        cfg.SOLVER.OPTIMIZER_CFG = dict(type="Ranger", lr=1e-4, weight_decay=0)
        model, opt = build_model_optimizer(cfg, is_test=False)
        model.load_state_dict({k: v.cuda() for k, v in synth.recipe_state_dict(expected_state_shapes(cfg)).items()}); model.train()
        b = {k: v.cuda() for k, v in synth.make_inputs(B, N, M, seed=3).items()}
        sym = [y_axis_symmetries(12) if i % 2 == 0 else None for i in range(B)]
        batch_updater_test(cfg, b)
        losses = []
        for it in range(3):
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                _, ld = model(b["x"], b["tfd_kps"], init_pose=b["obj_pose_est"], init_scale=b["obj_scale_est"], K_zoom=b["K"],
                              gt_ego_rot=b["gt_rot"], gt_trans=b["gt_trans"], gt_scale=b["gt_scale"], obj_kps=b["obj_kps"],
                              mean_scales=b["obj_mean_scales"], sym_info=sym, do_loss=True, cur_iter=1)
            tot = sum(ld.values()); tot.backward(); opt.step(); opt.zero_grad(set_to_none=True)
            losses.append(float(tot.detach()))
        torch.cuda.synchronize()
        ok = all(torch.isfinite(p).all() for p in model.parameters())
        print(B, N, M, "amp" if amp else "fp32", [round(l, 5) for l in losses], "finite params:", bool(ok))
