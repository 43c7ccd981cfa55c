"""Phase stamps of k_rot_l1_bwd_bf (autocast training; last tile of every workgroup).
    make -C catre_amd/csrc TRACE=1 && CATRE_HIP_LIB=$PWD/catre_amd/csrc/libcatre_hip_trace.so python profiles/trace_rot_l1_bwd_bf.py"""
import sys, os, torch, ctypes, logging
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.disable(logging.CRITICAL)
from catre_amd import hip, synth
from catre_amd.batching import batch_updater_test
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg
B, N, M = 256, 1024, 1024
cfg = default_cfg(device='cuda:0')
model, opt = build_model_optimizer(cfg, is_test=False)
model.load_state_dict({k: v.cuda() for k, v in synth.recipe_state_dict(expected_state_shapes(cfg)).items()}); model.train()
b = {k: v.cuda() for k, v in synth.make_inputs(B, N, M, seed=1).items()}
batch_updater_test(cfg, b)
def step():
    model.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        _, ld = model(b["x"], b["tfd_kps"], init_pose=b["obj_pose_est"], init_scale=b["obj_scale_est"], K_zoom=b["K"],
                      gt_ego_rot=b["gt_rot"], gt_trans=b["gt_trans"], gt_scale=b["gt_scale"], obj_kps=b["obj_kps"],
                      mean_scales=b["obj_mean_scales"], sym_info=[None] * B, do_loss=True, cur_iter=1)
    sum(ld.values()).backward()
step(); torch.cuda.synchronize()
OFF = 1 << 25
wgs = B * 1
buf = torch.zeros(OFF + wgs * 4 * 8 + 1024, dtype=torch.int64, device='cuda')
hip.load().catre_debug_trunk_trace(ctypes.c_void_p(buf.data_ptr()))
step(); torch.cuda.synchronize()
hip.load().catre_debug_trunk_trace(None)
t = buf[OFF:OFF + wgs * 32].view(wgs, 4, 8)[:, :, :6].cpu().double()
d = t[:, :, 1:] - t[:, :, :-1]
for i, nm in enumerate(['stage (loads + transform + LDS)', 'barrier', 'dA sweep + stores', 'dW sweep', 'barrier']):
    print(f'  {nm:34s} {d[:, :, i].mean():9.0f}  w0 {d[:, 0, i].mean():9.0f} w3 {d[:, 3, i].mean():9.0f}')
print('per tile', (t[:, :, 5] - t[:, :, 0]).mean().item(), ' (MFMA issue: dA 64 x 32 = 2048, dW 64 x 32 = 2048)')
