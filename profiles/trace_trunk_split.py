"""Phase stamps of k_trunk_split (s_memtime per wave): where a 64-point tile's cycles go."""
import sys, torch, ctypes
sys.path.insert(0, '/root/repo')
from catre_amd import hip, synth
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg
cfg = default_cfg(device='cuda:0')
model, _ = build_model_optimizer(cfg, is_test=True)
model.load_state_dict({k: v.cuda() for k, v in synth.recipe_state_dict(expected_state_shapes(cfg)).items()}); model.eval()
model.cfg.MODEL.CATRE.COMPUTE_DTYPE = "split"
B = 256
batch = {k: v.cuda() for k, v in synth.make_inputs(B, 1024, 1024, seed=1).items()}
model.refine(batch, n_iter=1)
tiles = B * 32
buf = torch.zeros(tiles * 8 * 8, dtype=torch.int64, device='cuda')
hip.load().catre_debug_trunk_trace(ctypes.c_void_p(buf.data_ptr()))
model.refine(batch, n_iter=1)
torch.cuda.synchronize()
hip.load().catre_debug_trunk_trace(None)
t = buf.view(tiles, 8, 8)[:, :8].cpu().double()
t = t[2048:6144]
d = t[:, :, 1:] - t[:, :, :-1]
names = ['P1 load+conv1+T64+bar', 'P2 ft mfma+bar', 'P3 pfmax/conv2+2bar', 'conv3 + split store', 'bar', 'conv4 pass a + max', 'conv4 pass b + max']
for i, nm in enumerate(names):
    print(f'  {nm:24s} {d[:, :, i].mean():10.0f}   wave0 {d[:, 0, i].mean():9.0f} wave7 {d[:, 7, i].mean():9.0f}')
tot = t[:, :, 7] - t[:, :, 0]
print('total per WG cycles', tot.mean().item(), ' max-wave', tot.max(1)[0].mean().item())
