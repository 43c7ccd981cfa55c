"""Does refining a B=256 batch as S independent sub-batches on S streams (objects are independent; the runtime keeps one
workspace per stream) hide the latency-bound launches (FC tails, reductions, heads) of one sub-batch under the MFMA kernels of
another?   python profiles/half_batch_streams.py [mode ...]"""
import json, logging, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.disable(logging.CRITICAL)
from catre_amd import synth
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg

B, N, M, K = 256, 1024, 1024, 4
cfg = default_cfg(num_pcl=N, num_kps=M, n_iter=K, device="cuda:0")
model, _ = build_model_optimizer(cfg, is_test=True)
model.load_state_dict({k: v.cuda() for k, v in synth.recipe_state_dict(expected_state_shapes(cfg)).items()})
model.eval()
batch = {k: v.cuda() for k, v in synth.make_inputs(B, N, M, seed=1000).items()}
for mode in (sys.argv[1:] or ["fp32", "split", "bf16"]):
    model.cfg.MODEL.CATRE.COMPUTE_DTYPE = mode
    ref = model.refine(batch, n_iter=K)[f"pose_{K}"].clone()
    for S in (1, 2, 4):
        subs = [{k: v[i * (B // S):(i + 1) * (B // S)].contiguous() for k, v in batch.items()} for i in range(S)]
        streams = [torch.cuda.Stream() for _ in range(S)]

        def run():
            outs = []
            cur = torch.cuda.current_stream()
            for st, sb in zip(streams, subs):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    outs.append(model.refine(sb, n_iter=K))
            for st in streams:
                cur.wait_stream(st)
            return outs
        for _ in range(3):
            outs = run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            outs = run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        got = torch.cat([o[f"pose_{K}"] for o in outs])
        print(json.dumps({"mode": mode, "sub_batches_on_streams": S, "ms_per_refine": round(dt * 1e3, 3),
                          "object_iterations_per_s": round(B * K / dt, 1), "bitwise_equal_to_one_batch": bool(torch.equal(got, ref))}))
