"""Bisect helper: refines of fixed batch sizes on 4 concurrent streams must reproduce the single-stream result.
`python profiles/soak_streams.py <mode> <rounds> <B...>`"""
import json, logging, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.disable(logging.CRITICAL)
from catre_amd import synth
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg

mode, rounds = sys.argv[1], int(sys.argv[2])
Bs = [int(a) for a in sys.argv[3:]]
N = M = 1024; K = 4
cfg = default_cfg(num_pcl=N, num_kps=M, n_iter=K, device="cuda:0")
model, _ = build_model_optimizer(cfg, is_test=True)
model.load_state_dict({k: v.cuda() for k, v in synth.recipe_state_dict(expected_state_shapes(cfg)).items()}); model.eval()
model.cfg.MODEL.CATRE.COMPUTE_DTYPE = mode
batches = {b: {k: v.cuda() for k, v in synth.make_inputs(b, N, M, seed=7 + b).items()} for b in Bs}
ref = {b: model.refine(batches[b], n_iter=K) for b in Bs}
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(4)]
bad, first = {}, {}
for r in range(rounds):
    outs = []
    for i, st in enumerate(streams):
        with torch.cuda.stream(st):
            b = Bs[(r + i) % len(Bs)]
            outs.append((b, model.refine(batches[b], n_iter=K)))
    torch.cuda.synchronize()
    for b, out in outs:
        for it in range(1, K + 1):
            if not torch.equal(out[f"pose_{it}"], ref[b][f"pose_{it}"]):
                bad[b] = bad.get(b, 0) + 1
                first[it] = first.get(it, 0) + 1
                break
print(json.dumps({"mode": mode, "Bs": Bs, "rounds": rounds, "mismatches_by_B": bad, "first_bad_iteration": first}))
