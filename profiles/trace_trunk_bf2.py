"""Phase stamps of k_trunk_bf2 (cycle counter per wave): where a 128-point pair's cycles go.
    make -C catre_amd/csrc TRACE=1 && CATRE_HIP_LIB=$PWD/catre_amd/csrc/libcatre_hip_trace.so python profiles/trace_trunk_bf2.py"""
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
pairs = B * 16
names = ['P1 load+conv1+T64+bar', 'P2 ft mfma+bar', 'P3 pfmax/conv2+2bar', 'conv3 + store', 'bar', 'conv4 pass a + max', 'conv4 pass b + max']
# ablations (argv: knob values; results are wrong with operands skipped - timing only): 1 = no weight loads, 2 = no LDS fragment loads
for knob in [int(a) for a in sys.argv[1:]] or [0]:
    hip.load().catre_debug_knob(1, knob)
    buf = torch.zeros(pairs * 8 * 8, dtype=torch.int64, device='cuda')
    hip.load().catre_debug_trunk_trace(ctypes.c_void_p(buf.data_ptr()))
    model.refine(batch, n_iter=1)
    torch.cuda.synchronize()
    hip.load().catre_debug_trunk_trace(None)
    t = buf.view(pairs, 8, 8).cpu().double()
    t = t[1024:3072]
    d = t[:, :, 1:] - t[:, :, :-1]
    print(f'--- ablation knob {knob}')
    for i, nm in enumerate(names):
        print(f'  {nm:24s} {d[:, :, i].mean():10.0f}   wave0 {d[:, 0, i].mean():9.0f} wave7 {d[:, 7, i].mean():9.0f}')
    tot = t[:, :, 7] - t[:, :, 0]
    print('total per WG cycles', tot.mean().item(), ' max-wave', tot.max(1)[0].mean().item())
hip.load().catre_debug_knob(1, 0)
print('ideal MFMA cycles per pair (4704 MFMAs x 32 cycles / 4 SIMDs):', 4704 * 32 / 4)
