"""Time the training GEMM ops alone at the training shapes (R = 256 x 2048 rows) and check them against torch:
    python profiles/gemm_probe.py          (env CATRE_PL_GRID=0/256/512 selects the pipelined row-GEMM grid)"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from catre_amd import train_ops as T

dev = "cuda:0"
R = 256 * 2048
g = torch.Generator().manual_seed(0)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


out = []
for (K, J, masked) in [(256, 256, False), (512, 128, True), (256, 64, False), (64, 256, False), (128, 64, True), (64, 128, False)]:
    x = torch.randn(R, K, generator=g).to(dev)
    w = (torch.randn(J, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(J, generator=g).to(dev)
    xm = torch.randn(R, K, generator=g).to(dev) if masked else None
    fn = lambda: T._gemm_nt(x, w, b, False, xmask=xm)
    y = fn()
    xr = (x * (xm > 0)) if masked else x
    ref = (xr[:4096].double() @ w.double().t() + b.double())
    err = float((y[:4096].double() - ref).abs().max())
    ref2 = (xr[-4096:].double() @ w.double().t() + b.double())
    err = max(err, float((y[-4096:].double() - ref2).abs().max()))
    us = timeit(fn)
    out.append({"op": "gemm_rows", "K": K, "J": J, "xmask": masked, "us": round(us, 1), "TFLOPs": round(2 * R * K * J / us / 1e6, 1),
                "GBps": round((R * K * (2 if masked else 1) + R * J) * 4 / us / 1e3, 0), "max_err": err})
    del x, xm, y
for (J, K) in [(256, 256), (512, 128), (256, 64), (128, 64), (64, 64), (64, 4)]:
    dy = torch.randn(R, J, generator=g).to(dev)
    x = torch.randn(R, K, generator=g).to(dev)
    fn = lambda: T._gemm_tn(dy, x, with_bias=True)
    dw, db = fn()
    ref = dy[:, :8].double().t() @ x.double()
    err = float((dw[:8].double() - ref).abs().max() / ref.abs().max())
    errb = float((db.double() - dy.double().sum(0)).abs().max())
    us = timeit(fn)
    out.append({"op": "gemm_tn", "J": J, "K": K, "us": round(us, 1), "TFLOPs": round(2 * R * K * J / us / 1e6, 1),
                "GBps": round(R * (J + K) * 4 / us / 1e3, 0), "rel_err": err, "db_err": errb})
    del dy, x
for o in out:
    print(json.dumps(o))
