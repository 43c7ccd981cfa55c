"""Training iteration as the victim: forward + device-side loss + backward of a B-object batch on one stream while a second
model refines in <aggressor mode> on another stream; losses and every parameter gradient must reproduce the solo run bit for
bit.  `python profiles/soak_train.py <aggressor mode> <rounds> [victim amp: fp32|bf16|split]`"""
import json, logging, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.disable(logging.CRITICAL)
from catre_amd import synth
from catre_amd.batching import batch_updater_test
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg

am, rounds = sys.argv[1], int(sys.argv[2])
vmode = sys.argv[3] if len(sys.argv) > 3 else "fp32"
N = M = 1024; B = 16
def mk(test):
    cfg = default_cfg(num_pcl=N, num_kps=M, n_iter=2, device="cuda:0")
    if not test and vmode != "fp32": cfg.MODEL.CATRE.COMPUTE_DTYPE = vmode
    m, _ = build_model_optimizer(cfg, is_test=test)
    m.load_state_dict({k: v.cuda() for k, v in synth.recipe_state_dict(expected_state_shapes(cfg)).items()})
    return m, cfg
vict, vcfg = mk(False); vict.train()
aggr, _ = mk(True); aggr.eval(); aggr.cfg.MODEL.CATRE.COMPUTE_DTYPE = am if am != "none" else "fp32"
b = {k: v.cuda() for k, v in synth.make_inputs(B, N, M, seed=4).items()}
ab = {k: v.cuda() for k, v in synth.make_inputs(16, N, M, seed=9).items()}
batch_updater_test(vcfg, b)
def step():
    vict.zero_grad(set_to_none=True)
    out, ld = vict(b["x"], b["tfd_kps"], init_pose=b["obj_pose_est"], init_scale=b["obj_scale_est"], K_zoom=b["K"],
                   gt_ego_rot=b["gt_rot"], gt_trans=b["gt_trans"], gt_scale=b["gt_scale"], obj_kps=b["obj_kps"],
                   mean_scales=b["obj_mean_scales"], sym_info=[None] * B, do_loss=True, cur_iter=1)
    sum(ld.values()).backward()
    g = {k: p.grad.clone() for k, p in vict.named_parameters() if p.grad is not None}
    g["__loss__"] = torch.stack([v.detach() for v in ld.values()])
    return g
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(sa): want = step()
with torch.no_grad(), torch.cuda.stream(sb): aggr.refine(ab, n_iter=2)
torch.cuda.synchronize()
bad, events = {}, []
for r in range(rounds):
    if am != "none":
        with torch.no_grad(), torch.cuda.stream(sb):
            for _ in range(3): aggr.refine(ab, n_iter=2)
    with torch.cuda.stream(sa): got = step()
    torch.cuda.synchronize()
    hit = [k for k in want if not torch.equal(got[k], want[k])]
    for k in hit: bad[k] = bad.get(k, 0) + 1
    if hit and len(events) < 4:
        events.append({"round": r, "n": len(hit), "same": sorted(k for k in want if k not in hit)[:12],
                       "max_rel": max(float(((got[k] - want[k]).abs().max() / (want[k].abs().max() + 1e-30))) for k in hit)})
print(json.dumps({"victim": f"training iteration ({vmode})", "aggressor": am, "rounds": rounds,
                  "n_tensors_that_differed": len(bad), "rounds_hit": max(bad.values()) if bad else 0,
                  "some": sorted(bad)[:6], "events": events}))
