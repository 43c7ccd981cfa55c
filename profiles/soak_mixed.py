"""Mixed-mode concurrency: a victim model refines batches of 2 / 12 objects in <victim mode> on one stream while a second
model refines in <aggressor mode> on another; the victim must reproduce its solo result bit for bit.
`python profiles/soak_mixed.py <victim mode> <aggressor mode> <rounds>`"""
import json, logging, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.disable(logging.CRITICAL)
from catre_amd import synth
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg

vm, am, rounds = sys.argv[1], sys.argv[2], int(sys.argv[3])
N = M = 1024; K = 2
def mk(mode):
    cfg = default_cfg(num_pcl=N, num_kps=M, n_iter=K, device="cuda:0")
    m, _ = build_model_optimizer(cfg, is_test=True)
    m.load_state_dict({k: v.cuda() for k, v in synth.recipe_state_dict(expected_state_shapes(cfg)).items()})
    m.cfg.MODEL.CATRE.COMPUTE_DTYPE = mode
    return m.eval()
vict, aggr = mk(vm), mk(am)
vb = {b: {k: v.cuda() for k, v in synth.make_inputs(b, N, M, seed=20 + b).items()} for b in (2, 12)}
ab = {k: v.cuda() for k, v in synth.make_inputs(16, N, M, seed=9).items()}
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
want = {}
with torch.cuda.stream(sa):
    for b in vb: want[b] = vict.refine(vb[b], n_iter=K)[f"pose_{K}"].clone()
with torch.cuda.stream(sb):
    aggr.refine(ab, n_iter=K)
torch.cuda.synchronize()
bad, runs = {}, 0
for r in range(rounds):
    with torch.cuda.stream(sb):
        aggr.refine(ab, n_iter=K); aggr.refine(ab, n_iter=K)
    with torch.cuda.stream(sa):
        outs = [(b, vict.refine(vb[b], n_iter=K)[f"pose_{K}"]) for b in (2, 12, 2, 12)]
    torch.cuda.synchronize()
    for b, o in outs:
        runs += 1
        if not torch.equal(o, want[b]): bad[b] = bad.get(b, 0) + 1
print(json.dumps({"victim": vm, "aggressor": am, "victim_refines": runs, "mismatches_by_B": bad}))
