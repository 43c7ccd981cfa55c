import sys, torch, ctypes
sys.path.insert(0, '/root/repo')
from catre_amd import hip, synth
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg
cfg = default_cfg(device='cuda:0')
model,_ = build_model_optimizer(cfg, is_test=True)
sd = synth.recipe_state_dict(expected_state_shapes(cfg))
model.load_state_dict({k:v.cuda() for k,v in sd.items()}); model.eval()
B=256
batch = {k:v.cuda() for k,v in synth.make_inputs(B,1024,1024,seed=1).items()}
model.refine(batch, n_iter=1)
tiles = B*32
# the rotation-head kernels of the instrumented build stamp at offset 1 << 24 of the same buffer
buf = torch.zeros((1 << 24) + tiles*8*16, dtype=torch.int64, device='cuda')
hip.load().catre_debug_trunk_trace(ctypes.c_void_p(buf.data_ptr()))
model.refine(batch, n_iter=1)
torch.cuda.synchronize()
hip.load().catre_debug_trunk_trace(None)
t = buf[:tiles*64].view(tiles,8,8).cpu().double()
# k_trunk4 (one wave per SIMD, CATRE_TRUNK4 != 0) stamps waves 0..3 only
W = 8 if (t[:,4:,0] != 0).any() else 4
t = t[:, :W]
# steady-state workgroups: skip first and last 1024
t = t[2048:6144]
d = t[:,:,1:] - t[:,:,:-1]
names=['P1 load+conv1+bar','P2 pf mfma+bar','P3 copy/max/conv2+bar','conv3+epi','bar','conv4','max epi']
print(f'{W} waves per workgroup; per-phase cycles mean over waves/WGs (first wave, last wave):')
for i,nm in enumerate(names):
    print(f'  {nm:24s} {d[:,:,i].mean():10.0f}   wave0 {d[:,0,i].mean():9.0f} wave{W-1} {d[:,W-1,i].mean():9.0f}')
tot = (t[:,:,7]-t[:,:,0])
print('total per WG cycles', tot.mean().item(), ' max-wave', tot.max(1)[0].mean().item())
