"""k_trunk4<true> (training forward of the trunk) launched back to back, outside a training iteration: is the ~280 us it takes
over the inference kernel its own work or the state the chip is in between the HBM-bound phases of an iteration?
    python profiles/trunk_save_probe.py"""
import json, logging, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.disable(logging.CRITICAL)
from catre_amd import hip, synth
from catre_amd.batching import batch_updater_test
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg

B, N, M = 256, 1024, 1024
cfg = default_cfg(device="cuda:0")
model, _ = build_model_optimizer(cfg, is_test=False)
model.load_state_dict({k: v.cuda() for k, v in synth.recipe_state_dict(expected_state_shapes(cfg)).items()})
b = {k: v.cuda() for k, v in synth.make_inputs(B, N, M, seed=3).items()}
batch_updater_test(cfg, b)
rt = model._runtime()
dev = b["x"].device
desc = hip.points_desc(b["x"], b["tfd_kps"])
buf = rt.train_encoder_buffers(B, N, M, dev)
trans3 = torch.eye(3, device=dev).reshape(1, 9).repeat(2 * B, 1).contiguous()
trans64 = torch.eye(64, device=dev).reshape(1, 4096).repeat(2 * B, 1).contiguous()
rt.train_stn3d(desc, buf, B, N, M, dev, 0)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


out = {"trunk_train_fwd_ms_back_to_back": round(timed(lambda: rt.train_trunk(desc, trans3, trans64, buf, B, N, M, dev, 0)), 4)}
model.eval()
with torch.no_grad():
    st = rt.stage_pointnet(b["x"], b["tfd_kps"])  # warm
    out["stage_pointnet_ms"] = round(timed(lambda: rt.stage_pointnet(b["x"], b["tfd_kps"]), 10), 4)
print(json.dumps(out))
