import sys, torch, logging
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.disable(logging.CRITICAL)
from catre_amd import synth, hip
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg
B, N, M = 256, 1024, 1024
cfg = default_cfg(num_pcl=N, num_kps=M, device="cuda:0")
model, _ = build_model_optimizer(cfg, is_test=False)
model.load_state_dict({k: v.cuda() for k, v in synth.recipe_state_dict(expected_state_shapes(cfg)).items()})
model.train()
b = {k: v.cuda() for k, v in synth.make_inputs(B, N, M, seed=2000).items()}
from catre_amd.batching import batch_updater_test
batch_updater_test(cfg, b)
rt = model._runtime()
dev = torch.device("cuda:0")
desc = hip.points_desc(b["x"], b["tfd_kps"])
buf = rt.train_encoder_buffers(B, N, M, dev)
rt.train_stn3d(desc, buf, B, N, M, dev)
# need real trans3/trans64: run the model forward to get them is complex; use identity-ish via actual tails
from catre_amd import train_forward as TF, train_ops as T
p = dict(model.named_parameters())
pts = torch.cat([TF._points_rows(b["x"]), TF._points_rows(b["tfd_kps"])], 0)
with torch.no_grad():
    g, pf, _ = TF.pointnet_rows_fused(pts, desc, rt, p, B, N, M)
torch.cuda.synchronize()
R = B * (N + M)
# re-run to fetch buffers: pointnet_rows_fused allocates its own buf; replicate to read indices
buf = rt.train_encoder_buffers(B, N, M, dev)
rt.train_stn3d(desc, buf, B, N, M, dev)
trans = TF._stn(pts, p, "pcl_net.stn", 3, B, N, M, pre=(buf["a1"], buf["a2"], buf["g_stn"], buf["i_stn"]))
trans3 = trans.detach().reshape(-1, 9).contiguous()
rt.train_stnkd(desc, trans3, buf, B, N, M, dev)
h1 = buf["h1"]
tf_ = TF._stn(h1, p, "pcl_net.fstn", 64, B, N, M, pre=(buf["f1"], buf["f2"], buf["g_fstn"], buf["i_fstn"]))
trans64 = tf_.detach().reshape(-1, 4096).contiguous()
rt.train_trunk(desc, trans3, trans64, buf, B, N, M, dev)
torch.cuda.synchronize()
for k in ("i_stn", "i_fstn", "i"):
    idx = buf[k].long()
    mark = torch.zeros(R, dtype=torch.bool, device=dev)
    mark[idx.reshape(-1)] = True
    print(k, "fraction of rows that are an arg-max of some channel:", float(mark.float().mean()))
