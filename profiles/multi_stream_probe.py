"""Several images refined concurrently: S streams, each running K=4 refines of its own B-object batch back to back.
A single small batch cannot fill 256 CUs (B=1: 32-256 workgroups per kernel, most kernels a few microseconds), so
independent calls on separate streams overlap.  `python profiles/multi_stream_probe.py [B] [streams...]`"""
import json, logging, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.disable(logging.CRITICAL)
from catre_amd import synth
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
counts = [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8]
N = M = 1024; K = 4
cfg = default_cfg(num_pcl=N, num_kps=M, n_iter=K, device="cuda:0")
model, _ = build_model_optimizer(cfg, is_test=True)
model.load_state_dict({k: v.cuda() for k, v in synth.recipe_state_dict(expected_state_shapes(cfg)).items()}); model.eval()
batches = [{k: v.cuda() for k, v in synth.make_inputs(B, N, M, seed=40 + i).items()} for i in range(max(counts))]
want = [model.refine(b, n_iter=K)[f"pose_{K}"].clone() for b in batches]
torch.cuda.synchronize()
for S in counts:
    streams = [torch.cuda.Stream() for _ in range(S)]
    def sweep(reps):
        outs = [None] * S
        for _ in range(reps):
            for i, st in enumerate(streams):
                with torch.cuda.stream(st):
                    outs[i] = model.refine(batches[i], n_iter=K)
        return outs
    sweep(5); torch.cuda.synchronize()
    reps = 100
    t0 = time.perf_counter(); outs = sweep(reps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    same = all(torch.equal(o[f"pose_{K}"], w) for o, w in zip(outs, want))
    print(json.dumps({"B": B, "streams": S, "refines_per_s": round(S * reps / dt, 1),
                      "ms_per_refine_amortised": round(dt / (S * reps) * 1e3, 4), "results_equal_single_stream": same}))
    # the same with one captured graph per stream (catre_amd.graphed.GraphedRefine): the host launches one graph per refine
    from catre_amd.graphed import GraphedRefine
    graphs = [GraphedRefine(model, batches[i], n_iter=K) for i in range(S)]
    def gsweep(reps):
        outs = [None] * S
        for _ in range(reps):
            for i, st in enumerate(streams):
                with torch.cuda.stream(st):
                    outs[i] = graphs[i](batches[i])
        return outs
    gsweep(5); torch.cuda.synchronize()
    t0 = time.perf_counter(); outs = gsweep(reps); t1 = time.perf_counter(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    same = all(torch.equal(o[f"pose_{K}"], w) for o, w in zip(outs, want))
    print(json.dumps({"B": B, "streams": S, "graphed": True, "refines_per_s": round(S * reps / dt, 1),
                      "ms_per_refine_amortised": round(dt / (S * reps) * 1e3, 4),
                      "host_ms_per_refine": round((t1 - t0) / (S * reps) * 1e3, 4), "results_equal_single_stream": same}))
    del graphs
