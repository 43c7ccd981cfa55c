"""Is there anything to win from running the NEXT iteration's observed-cloud STN3d beside the rotation heads of the current
one (the observed cloud of iteration i+1 only needs the translation, which the ts head has already produced)?  Times the
rot-head chain and an observed-only STN3d stage back to back on one stream and side by side on two (fp32 stage entry points,
separate workspaces), at the headline shape.   python profiles/overlap_probe.py"""
import ctypes, json, logging, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.disable(logging.CRITICAL)
from catre_amd import hip, synth
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg

B, N, M = 256, 1024, 1024
cfg = default_cfg(num_pcl=N, num_kps=M, n_iter=4, device="cuda:0")
model, _ = build_model_optimizer(cfg, is_test=True)
model.load_state_dict({k: v.cuda() for k, v in synth.recipe_state_dict(expected_state_shapes(cfg)).items()})
model.eval()
batch = {k: v.cuda() for k, v in synth.make_inputs(B, N, M, seed=1000).items()}
model.refine(batch, n_iter=1)
rt = model._runtime()
lib = hip.load()
dev = torch.device("cuda:0")
x = torch.randn(B, 3, N, device=dev)
kp = torch.randn(B, 3, M, device=dev)
st = rt.stage_pointnet(x, kp)
gfeat, pointfeat = st["gfeat"], st["pointfeat"]
prm, packed = rt.params(dev)
pts = hip.points_desc(x, x)
side = torch.cuda.Stream(priority=0)
low = torch.cuda.Stream(priority=1 if False else 0)
main = torch.cuda.current_stream()
ws_main = rt.workspace(B, N, M, dev)
with torch.cuda.stream(side):
    ws_side = rt.workspace(B, N, M, dev)
rot = torch.empty(B, 6, device=dev)
pool = torch.empty(B, 1024, device=dev)


def heads(stream):
    hip.check(lib.catre_rot_head_dim(hip.ptr(gfeat), hip.ptr(pointfeat), prm, hip.ptr(packed), hip.ptr(rot), hip.ptr(ws_main),
                                     ws_main.numel(), B, N, M, 3, stream.cuda_stream), "rot")


def stn(stream):
    hip.check(lib.catre_stn3d_pool(ctypes.byref(pts), prm, hip.ptr(packed), hip.ptr(pool), hip.ptr(ws_side), ws_side.numel(),
                                   B, N, 0, stream.cuda_stream), "stn")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    for _ in range(reps):
        fn()
    e1.record(main)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def serial():
    heads(main)
    stn(main)


def overlapped():
    side.wait_stream(main)
    stn(side)
    heads(main)
    main.wait_stream(side)


def overlapped_late():  # the side stream starts after the heads were enqueued (dispatch order within the queues is the same)
    side.wait_stream(main)
    heads(main)
    stn(side)
    main.wait_stream(side)


res = {"heads_us": round(timed(lambda: heads(main)), 1), "stn3d_obs_us": round(timed(lambda: stn(main)), 1),
       "serial_us": round(timed(serial), 1), "two_streams_us": round(timed(overlapped), 1),
       "two_streams_heads_first_us": round(timed(overlapped_late), 1)}
res["gain_us"] = round(res["serial_us"] - min(res["two_streams_us"], res["two_streams_heads_first_us"]), 1)
print(json.dumps(res))
