"""Soak test of the small-batch latency path: repeated refines of batches of 1..8 objects (fp32 and split), alone and on
four concurrent streams, must reproduce the first result bit for bit every time (no race in the workgroup-role kernels,
the early-exiting waves, the per-stream workspaces).  `python profiles/soak_small.py [rounds=300]`"""
import json, logging, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.disable(logging.CRITICAL)
from catre_amd import synth
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
N = M = 1024; K = 4
cfg = default_cfg(num_pcl=N, num_kps=M, n_iter=K, device="cuda:0")
model, _ = build_model_optimizer(cfg, is_test=True)
model.load_state_dict({k: v.cuda() for k, v in synth.recipe_state_dict(expected_state_shapes(cfg)).items()}); model.eval()
big = {k: v.cuda() for k, v in synth.make_inputs(12, N, M, seed=7).items()}
res = {}
for mode in ("fp32", "split"):
    model.cfg.MODEL.CATRE.COMPUTE_DTYPE = mode
    ref = model.refine(big, n_iter=K)
    subs = {b: {k: v[:b].contiguous() for k, v in big.items()} for b in range(1, 9)}
    bad = 0
    where = {}
    t0 = time.perf_counter()
    for r in range(rounds):
        for b in range(1, 9):
            out = model.refine(subs[b], n_iter=K)
            if not (torch.equal(out[f"pose_{K}"], ref[f"pose_{K}"][:b]) and torch.equal(out[f"scale_{K}"], ref[f"scale_{K}"][:b])):
                bad += 1
                where[f"seq_b{b}"] = where.get(f"seq_b{b}", 0) + 1
    streams = [torch.cuda.Stream() for _ in range(4)]
    for r in range(rounds):
        outs = []
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                b = 1 + (r + i) % 8
                outs.append((b, model.refine(subs[b], n_iter=K)))
        torch.cuda.synchronize()
        for b, out in outs:
            if not torch.equal(out[f"pose_{K}"], ref[f"pose_{K}"][:b]):
                bad += 1
                where[f"streams_b{b}"] = where.get(f"streams_b{b}", 0) + 1
    res[mode] = {"refines": rounds * 8 + rounds * 4, "mismatches": bad, "where": where, "seconds": round(time.perf_counter() - t0, 1)}
print(json.dumps(res))
