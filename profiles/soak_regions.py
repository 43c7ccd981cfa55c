"""Which kernel misbehaves under concurrency: one refine iteration per stream on 4 streams, then every region of each
stream's workspace is compared with the same stream's solo run.  `python profiles/soak_regions.py <mode> <B> <rounds>`"""
import json, logging, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.disable(logging.CRITICAL)
from catre_amd import synth
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg

mode, B, rounds = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
N = M = 1024; PMW = 1088
cfg = default_cfg(num_pcl=N, num_kps=M, n_iter=1, device="cuda:0")
model, _ = build_model_optimizer(cfg, is_test=True)
model.load_state_dict({k: v.cuda() for k, v in synth.recipe_state_dict(expected_state_shapes(cfg)).items()}); model.eval()
model.cfg.MODEL.CATRE.COMPUTE_DTYPE = mode

def layout(b):
    T, P = N // 64 + M // 64, N + M
    sizes = [("tspart", b * 8 * 256), ("xbuf", b * N * 3), ("kbuf", b * M * 3), ("pm", (2 if 2 * b <= 16 else 1) * b * T * PMW),
             ("pool", 2 * b * 1024), ("h1", 2 * b * 512), ("h2", 2 * b * 256), ("trans3", 2 * b * 9), ("trans64", 2 * b * 4096),
             ("gfeat", 2 * b * PMW), ("pointfeat", b * P * 64), ("dt", b * 3), ("ds", b * 3), ("rot6d", b * 6),
             ("bias0", 4 * b * 256), ("gn0", b * 2 * T * 64), ("gn1", b * 2 * T * 64), ("aff0", b * 8 * 256),
             ("gn1stat", b * 2 * 64), ("y1", max(b * 2 * P * 256, b * 2 * (4 * (4096 + 64) + 64))), ("rpart", b * 2 * T * 4)]
    out, o = {}, 0
    for k, n in sizes:
        out[k] = (o, n); o += (n + 63) // 64 * 64
    return out

L = layout(B)
order = ["trans3", "trans64", "gfeat", "pointfeat", "tspart", "dt", "bias0", "aff0", "gn1", "gn1stat", "y1", "rpart"]
rt = model._runtime()
streams = [torch.cuda.Stream() for _ in range(4)]
batches = [{k: v.cuda() for k, v in synth.make_inputs(B, N, M, seed=50 + i).items()} for i in range(4)]
refs = []
for i, st in enumerate(streams):
    with torch.cuda.stream(st):
        model.refine(batches[i], n_iter=1)
    torch.cuda.synchronize()
    ws = rt._ws[(0, st.cuda_stream)].view(torch.float32)
    refs.append(ws.clone())
first = {}
for r in range(rounds):
    for i, st in enumerate(streams):
        with torch.cuda.stream(st):
            model.refine(batches[i], n_iter=1)
    torch.cuda.synchronize()
    for i, st in enumerate(streams):
        ws = rt._ws[(0, st.cuda_stream)].view(torch.float32)
        for k in order:
            o, n = L[k]
            a, b_ = ws[o:o + n], refs[i][o:o + n]
            if not torch.equal(a.view(torch.int32), b_.view(torch.int32)):
                d = (a - b_).abs()
                d = d[torch.isfinite(d)]
                first.setdefault(k, [0, 0.0])
                first[k][0] += 1
                first[k][1] = max(first[k][1], float(d.max()) if d.numel() else -1.0)
                if k in ("y1", "rpart") and len(first[k]) < 5:
                    idx = torch.nonzero(a.view(torch.int32) != b_.view(torch.int32)).flatten()
                    first[k].append({"n_diff": int(idx.numel()), "first": int(idx[0]), "last": int(idx[-1]),
                                     "idx": idx[:12].tolist(), "got": a[idx[:6]].tolist(), "want": b_[idx[:6]].tolist()})
                break
print(json.dumps({"mode": mode, "B": B, "rounds": rounds, "first_region_that_differs": first}))
