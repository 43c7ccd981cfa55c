"""Phase stamps of k_rot_l1_bf (cycle counter per wave).
    make -C catre_amd/csrc TRACE=1 && CATRE_HIP_LIB=$PWD/catre_amd/csrc/libcatre_hip_trace.so python profiles/trace_rot_bf.py"""
import sys, os, torch, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from catre_amd import hip, synth
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg
cfg = default_cfg(device='cuda:0')
model, _ = build_model_optimizer(cfg, is_test=True)
model.load_state_dict({k: v.cuda() for k, v in synth.recipe_state_dict(expected_state_shapes(cfg)).items()}); model.eval()
model.cfg.MODEL.CATRE.COMPUTE_DTYPE = "bf16"
B = 256
batch = {k: v.cuda() for k, v in synth.make_inputs(B, 1024, 1024, seed=1).items()}
model.refine(batch, n_iter=1)
tiles = B * 32
OFF = 1 << 24
buf = torch.zeros(OFF + tiles * 8 * 16, dtype=torch.int64, device='cuda')
hip.load().catre_debug_trunk_trace(ctypes.c_void_p(buf.data_ptr()))
model.refine(batch, n_iter=1)
torch.cuda.synchronize()
hip.load().catre_debug_trunk_trace(None)
t = buf[OFF:].view(tiles, 8, 16)[:, :4, :14].cpu().double()
t = t[2048:6144]
d = t[:, :, 1:] - t[:, :, :-1]
names = ['load pf+bar', 'h0 layer0 mfma', 'h0 gelu+lds', 'h0 bar', 'h0 layer1 mfma', 'h0 store+stats', 'h0 bar',
         'h1 layer0', 'h1 gelu', 'h1 bar', 'h1 layer1', 'h1 store+stats', 'h1 bar']
for i, nm in enumerate(names):
    print(f'  {nm:18s} {d[:, :, i].mean():9.0f}  w0 {d[:, 0, i].mean():9.0f} w3 {d[:, 3, i].mean():9.0f}')
print('total per WG', (t[:, :, 13] - t[:, :, 0]).mean().item())
span = (t[:, :, 13].max() - t[:, :, 0].min()).item()
print('span of the sampled WGs (cycles)', span, ' concurrent WGs per CU ~', (t[:, 0, 13] - t[:, 0, 0]).sum().item() / span / 256)
