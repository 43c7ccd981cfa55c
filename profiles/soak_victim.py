"""Victim / aggressor probe: the fp32 rot-head stage chain (k_linear, k_pf_moments, k_gn0_from_moments, k_rot_l1,
k_gn_finalize, k_rot_out, k_rot_finish) runs on one stream while another stream runs refines in <mode>; the victim's
output must stay bit-identical to its solo result.  `python profiles/soak_victim.py <aggressor mode> <rounds>`"""
import json, logging, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.disable(logging.CRITICAL)
from catre_amd import synth
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg

mode, rounds = sys.argv[1], int(sys.argv[2])
N = M = 1024
def mk():
    cfg = default_cfg(num_pcl=N, num_kps=M, n_iter=2, device="cuda:0")
    m, _ = build_model_optimizer(cfg, is_test=True)
    m.load_state_dict({k: v.cuda() for k, v in synth.recipe_state_dict(expected_state_shapes(cfg)).items()})
    return m.eval()
vict, aggr = mk(), mk()
aggr.cfg.MODEL.CATRE.COMPUTE_DTYPE = mode
B = 12
gen = torch.Generator().manual_seed(3)
gfeat = torch.randn(2 * B, 1088, generator=gen).cuda()
pointfeat = torch.randn(B * (N + M), 64, generator=gen).cuda()
rt = vict._runtime()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(sa):
    want = rt.stage_rot_head(gfeat, pointfeat, B, N, M).clone()
abatch = {k: v.cuda() for k, v in synth.make_inputs(16, N, M, seed=9).items()}
with torch.cuda.stream(sb):
    aggr.refine(abatch, n_iter=2)
torch.cuda.synchronize()
bad, cols = 0, {}
for r in range(rounds):
    with torch.cuda.stream(sb):
        aggr.refine(abatch, n_iter=2)
    with torch.cuda.stream(sa):
        outs = [rt.stage_rot_head(gfeat, pointfeat, B, N, M).clone() for _ in range(4)]
    torch.cuda.synchronize()
    for o in outs:
        if not torch.equal(o, want):
            bad += 1
            for j in torch.nonzero((o != want).any(0)).flatten().tolist():
                cols[j] = cols.get(j, 0) + 1
print(json.dumps({"aggressor": mode, "victim": "fp32 rot-head stage chain", "runs": rounds * 4, "mismatches": bad,
                  "rot6d_columns_hit": cols}))
