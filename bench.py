#!/usr/bin/env python
"""bench.py - pose-refine throughput of the MI355X-native CATRE hot path.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch = a K_ITER=4 refine (pose-apply -> shared PointNet
on observed cloud + transformed prior -> t/s head + rotation heads -> SO(3)/scale update, looped 4x on
the device) of B=256 objects with N=M=1024 points per rank: 1024 object-iterations per rank-step
(BASELINE.json metric "pose-refine iters/sec (B=256, N=1024, K=4)").  Inputs are synthetic (seeded,
reference shapes), weights are the seeded recipe; both are resident in HBM before the timed region.

Multi-GPU: objects are independent, so ranks hold disjoint batches and the data path
has no collective ("weak" scaling: 256 objects per GPU).  Timing: barrier + synchronize, K steps,
synchronize + barrier, MAX over ranks.

The JSON line also carries
  roofline     - the dominant kernel (k_trunk4: feature transform + conv2..conv4 + max-pool, 57 % of the
                 FLOPs) measured live with HIP events recorded on its launch stream inside the timed
                 region, against the dense fp32 MFMA peak (157.3 TFLOP/s);
  cpu_baseline - the oracle (torch CPU port of the reference path) timed on this box's host cores on a
                 bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

B_PER_GPU, N_PTS, M_PTS, K_ITER = 256, 1024, 1024, 4
FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same guide: BF16 dense (no sparsity)

# algorithmic FLOPs: per point of one cloud through k_trunk4 / k_trunk
#   pointfeat = h1^T T64 (2*64*64) + conv2 (2*64*128) + conv3 (2*128*512) + conv4 (2*512*1024) + conv1/T3 (2*3*64 + 18)
TRUNK_FLOPS_PER_POINT = 2 * (64 * 64 + 64 * 128 + 128 * 512 + 512 * 1024) + 2 * 3 * 64 + 18


def flops_per_object_iteration(N, M):
    """SURVEY.md 8d general form (rot-head layer 0 counted in its restructured form)."""
    return 1770258 * (N + M) + 9446400 + 2 * ((2 * 64 * 256 + 2 * 256 * 256 + 2 * 256 * 3 + 6) * (N + M) + 4 * 1024 * 256) + 692736


# CATRE_BENCH_DRYRUN=1: no GPU work - the launcher, rendezvous (gloo on CPU), barrier / MAX-over-ranks timing and
# rank-0 JSON plumbing only (tests/test_bench_launcher.py runs `python bench.py --gpus 2` this way).  The line it
# prints is marked "dryrun" and carries no throughput.
DRYRUN = os.environ.get("CATRE_BENCH_DRYRUN", "0") == "1"
# CATRE_BENCH_SHARE_GPU=1: every rank of an N > 1 job drives GPU 0 and the job's collectives go through gloo (RCCL refuses
# two ranks on one device).  For boxes with ONE GPU: the launcher, the rendezvous, DDP's gradient exchange and the timing /
# JSON plumbing run with real device work (tests/test_multi_gpu.py).  The line is marked "shared_gpu": true - its `value`
# is NOT a scaling number.
SHARE_GPU = os.environ.get("CATRE_BENCH_SHARE_GPU", "0") == "1"
GRAD_ALLREDUCE_BYTES = 4297175 * 4  # trainable-and-used fp32 parameters: one all-reduce per backward


_OUT = [None]


def claim_stdout():
    """The driver reads ONE JSON line from stdout.  Libraries write there too (RCCL prints a version banner under
    NCCL_DEBUG=VERSION, which the GPU boxes export): keep the real stdout for the line and point fd 1 at stderr for everyone
    else - this process's C libraries and its children."""
    if _OUT[0] is None:
        sys.stdout.flush()
        _OUT[0] = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit(line):
    out = _OUT[0] or sys.stdout
    out.write(json.dumps(line) + "\n")
    out.flush()


def self_launch(n):
    """Re-run this script as `n` ranks under torch.distributed.run on 127.0.0.1 (free port) and return its exit code."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    return subprocess.run(cmd).returncode   # (the ranks inherit the real stdout; rank 0 writes the line, everything else goes to stderr)


def rank_stats(dist, dev, dt):
    """(max dt over ranks, per-rank ms list, number of ranks that took part) - one all_gather on the job's backend."""
    if dist is None:
        return dt, [round(dt * 1e3, 3)], 1
    mine = torch.tensor([dt, 1.0], device="cpu" if SHARE_GPU else dev, dtype=torch.float64)
    allv = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(allv, mine)
    allv = torch.stack(allv).cpu()
    return float(allv[:, 0].max()), [round(float(v) * 1e3, 3) for v in allv[:, 0]], int(allv[:, 1].sum())


def comm_info(dist, dev, world, rank, local_rank):
    """What a reader of an N > 1 line needs to trust it: the collective library and its version, and one
    `NCCL_DEBUG`-style line per rank (rank -> device, PCI bus id, which peers it reaches over P2P / xGMI), gathered
    through the job's own backend."""
    info = {"backend": "none (single rank)" if dist is None else dist.get_backend()}
    try:
        v = torch.cuda.nccl.version()
        info["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception as e:  # noqa: BLE001 - a CPU-only torch build has no nccl module
        info["rccl_version"] = f"unavailable ({type(e).__name__})"
    info["hip"] = torch.version.hip
    props = torch.cuda.get_device_properties(dev)
    ndev = torch.cuda.device_count()
    peers = [j for j in range(ndev) if j != dev.index and torch.cuda.can_device_access_peer(dev.index, j)]
    mine = {"rank": rank, "local_rank": local_rank, "device": props.name, "gcn_arch": getattr(props, "gcnArchName", "?"),
            "pci_bus_id": getattr(props, "pci_bus_id", None), "visible_devices": ndev, "p2p_peers": peers}
    if dist is None:
        info["ranks"] = [mine]
    else:
        allv = [None] * world
        dist.all_gather_object(allv, mine)
        info["ranks"] = allv
    for k in ("NCCL_DEBUG", "NCCL_ALGO", "NCCL_PROTO", "RCCL_MSCCL_ENABLE", "HSA_ENABLE_IPC_MODE_LEGACY"):
        if k in os.environ:
            info.setdefault("env", {})[k] = os.environ[k]
    return info


def bench_dryrun(args, world, rank):
    import torch.distributed as dist

    dev = torch.device("cpu")
    if world > 1:
        dist.init_process_group(backend="gloo")
    else:
        dist = None

    def barrier():
        if dist is not None:
            dist.barrier()

    x = torch.randn(64, 64)
    for _ in range(args.warmup):
        x = torch.tanh(x @ x)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x = torch.tanh(x @ x)
        if args.mode == "train" and dist is not None:
            g = torch.zeros(1024)
            dist.all_reduce(g)  # stands in for DDP's gradient all-reduce
    barrier()
    dt, per_rank, seen = rank_stats(dist, dev, time.perf_counter() - t0)
    if rank == 0:
        emit(({
            "metric": "DRYRUN (launcher / rendezvous plumbing only, no GPU work)", "value": None, "unit": "object-iterations/s",
            "n_gpus": world, "ranks_seen": seen, "per_rank_ms": per_rank, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "dryrun", "dryrun": True, "mode": args.mode,
            "allreduce_bytes_per_step": GRAD_ALLREDUCE_BYTES * K_ITER if (args.mode == "train" and world > 1) else 0,
            "config": {"workload": "dryrun", "backend": "gloo"},
        }))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def train_flops_per_iteration():
    """MFMA work of one fp32 training iteration (forward + loss + backward + step at B=256, N=M=1024) as COUNTED by the
    committed PMC pass of the newest round (profiles/rNN_train_pmc_summary.csv: sum over kernels of dispatches x
    SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 FLOP, per iteration) -> (FLOP, source file) or (None, None)."""
    import glob
    import importlib.util

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_train_pmc_summary.csv")))
    if not files:
        return None, None
    spec = importlib.util.spec_from_file_location("make_tables", os.path.join(ROOT, "profiles", "make_tables.py"))
    mt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mt)
    pm = mt._raw_pmc(files[-1])
    its = sum(r["n"] for r in pm if r["k"] in ("k_trunk4<true>", "k_trunk<1, true>"))   # once per iteration, either form
    if not its:
        return None, None
    return sum(r["n"] * r["flop"] for r in pm) / its, os.path.relpath(files[-1], ROOT)


def cpu_baseline(cfg_fn, sd):
    """Oracle (port of the reference's CPU path) on the host cores, bounded sample."""
    from catre_amd import synth
    from oracle import catre_oracle as O

    cores = os.cpu_count() or 1
    cfg = cfg_fn("cpu")
    # torch's intra-op pool over-subscribes badly on many-core hosts for these small ops: calibrate the
    # thread count on a tiny sample first (the reference would run with whatever OMP_NUM_THREADS gives it)
    cal = synth.make_inputs(4, N_PTS, M_PTS, seed=122)
    best, best_t = None, None
    with torch.no_grad():
        for th in sorted({min(cores, t) for t in (8, 16, 32, 64, 128)}):
            torch.set_num_threads(th)
            O.refine_k({k: v[:1] for k, v in cal.items()}, sd, cfg, n_iter=1)
            t0 = time.perf_counter()
            O.refine_k(cal, sd, cfg, n_iter=1)
            t = time.perf_counter() - t0
            if best_t is None or t < best_t:
                best, best_t = th, t
        torch.set_num_threads(best)
        # bounded sample: three runs of ~6 s of CPU work each (best_t is 4 object-iterations); the median is reported
        Ks = 4
        Bs = int(max(2, min(64, 6.0 / max(best_t / 4.0, 1e-3) / Ks)))
        batch = synth.make_inputs(Bs, N_PTS, M_PTS, seed=123)
        runs = []
        for _ in range(3):
            t0 = time.perf_counter()
            O.refine_k(batch, sd, cfg, n_iter=Ks)
            runs.append(time.perf_counter() - t0)
    dt = sorted(runs)[1]
    # BASELINE config 1 (the reference's own CPU-runnable case: one object per call, catre_evaluator.py:292-311): B = 1
    # with ITS OWN thread-count calibration (the count that is best for a batch over-subscribes a single object: round 5's
    # driver line ran B = 1 on the batch's 32 threads and got 42 object-iterations/s), median of five K = 4 refines
    one = synth.make_inputs(1, N_PTS, M_PTS, seed=124)
    batch_threads = torch.get_num_threads()
    with torch.no_grad():
        b1, b1_t = None, None
        for th in sorted({min(cores, t) for t in (4, 8, 16, 32, 64)}):
            torch.set_num_threads(th)
            O.refine_k(one, sd, cfg, n_iter=1)
            t = None
            for _ in range(3):  # best of three: one 2-iteration trial per count let a noisy neighbour pick 32 threads (85 ms
                t0 = time.perf_counter()  # per refine) where 16 take 35 ms
                O.refine_k(one, sd, cfg, n_iter=2)
                dt_ = time.perf_counter() - t0
                t = dt_ if t is None or dt_ < t else t
            if b1_t is None or t < b1_t:
                b1, b1_t = th, t
        torch.set_num_threads(b1)
        O.refine_k(one, sd, cfg, n_iter=1)
        r1 = []
        for _ in range(5):
            t0 = time.perf_counter()
            O.refine_k(one, sd, cfg, n_iter=Ks)
            r1.append(time.perf_counter() - t0)
    dt1 = sorted(r1)[2]
    b1_threads = torch.get_num_threads()
    torch.set_num_threads(batch_threads)
    phys = None
    try:  # physical cores = distinct (physical id, core id) pairs
        ids, cur = set(), {}
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ":" in ln:
                    k, v = (t.strip() for t in ln.split(":", 1))
                    cur[k] = v
                elif cur:
                    ids.add((cur.get("physical id"), cur.get("core id")))
                    cur = {}
        if cur:
            ids.add((cur.get("physical id"), cur.get("core id")))
        phys = len(ids) if ids and (None, None) not in ids else None
    except OSError:
        pass
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {
        "value": round(Bs * Ks / dt, 3),
        "unit": "object-iterations/s",
        "cores": torch.get_num_threads(),
        "host_threads": cores,
        "physical_cores": phys,
        "kind": "port",
        "cpu_model": model,
        "config1_B1": {"value": round(Ks / dt1, 3), "unit": "object-iterations/s", "ms_per_refine": round(dt1 * 1e3, 1),
                       "cores": b1_threads,
                       "sample": f"B=1, N=M={N_PTS}, K={Ks}, median of 5 refines, {b1_threads} threads (calibrated for B=1)"},
        "runs_s": [round(r, 2) for r in runs],
        "sample": f"oracle/catre_oracle.refine_k (torch CPU fp32), B={Bs}, N=M={N_PTS}, K={Ks}, median of 3 runs "
                  f"({dt:.1f} s), {torch.get_num_threads()} of {cores} host threads (best of a thread-count calibration) on {model}",
    }


def ddp_bucket_plan(named_params, bucket_cap_mb=25):
    """The buckets `DistributedDataParallel(find_unused_parameters=True)` builds and keeps (torch/nn/parallel/distributed.py
    `_ddp_init_helper`: sizes [1 MiB, bucket_cap] over the parameters in definition order, bucket list reversed; with
    find_unused_parameters the reducer never rebuilds them) -> list of lists of parameter names, in the order the reducer
    launches their all-reduces."""
    import torch.distributed as tdist

    named = [(k, p) for k, p in named_params if p.requires_grad]
    idx, _ = tdist._compute_bucket_assignment_by_size([p for _, p in named],
                                                      [tdist._DEFAULT_FIRST_BUCKET_BYTES, int(bucket_cap_mb * 1024 * 1024)])
    return [[named[i][0] for i in b] for b in reversed(idx)]


class GradOrderProbe:
    """Records, for ONE backward, the order in which the parameters' gradients are accumulated and a device event behind each
    (post-accumulate-grad hooks: the point where DDP's reducer marks a gradient ready).  `report` relates that order to a
    bucket plan: a bucket's all-reduce can start once the LAST of its gradients is in."""

    def __init__(self, named_params, use_events=True):
        self.names, self.events, self.handles = [], [], []
        self.use_events = use_events
        for k, p in named_params:
            if p.requires_grad:
                self.handles.append(p.register_post_accumulate_grad_hook(self._hook(k)))
        self.t_begin = self.t_end = None

    def _hook(self, k):
        def fn(_p):
            self.names.append(k)
            if self.use_events:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                self.events.append(e)
        return fn

    def begin(self):
        self.names, self.events = [], []
        if self.use_events:
            self.t_begin = torch.cuda.Event(enable_timing=True)
            self.t_begin.record()

    def end(self):
        if self.use_events:
            self.t_end = torch.cuda.Event(enable_timing=True)
            self.t_end.record()

    def remove(self):
        for h in self.handles:
            h.remove()

    def report(self, plan):
        pos = {k: i for i, k in enumerate(self.names)}
        n = len(self.names)
        total_ms = self.t_begin.elapsed_time(self.t_end) if self.use_events else None
        out = []
        for b, names in enumerate(plan):
            used = [k for k in names if k in pos]
            if not used:
                out.append({"bucket": b, "tensors": len(names), "unused_only": True})
                continue
            last = max(used, key=lambda k: pos[k])
            first = min(used, key=lambda k: pos[k])
            row = {"bucket": b, "tensors": len(names), "tensors_with_grad": len(used),
                   "first_grad": first, "closes_with": last, "closes_at_hook": pos[last] + 1, "of_hooks": n}
            if self.use_events:
                ms = self.t_begin.elapsed_time(self.events[pos[last]])
                row["closes_ms_into_backward"] = round(ms, 3)
                row["backward_ms"] = round(total_ms, 3)
                row["backward_left_to_hide_allreduce_ms"] = round(total_ms - ms, 3)
            out.append(row)
        return out


def init_world1_group(dev):
    """A one-rank RCCL process group on 127.0.0.1 (free port): what `DistributedDataParallel` needs to run its reducer -
    hooks, find_unused_parameters graph walk, bucket copies, one all-reduce launch per bucket - on a single GPU."""
    import socket

    import torch.distributed as tdist

    if not tdist.is_initialized():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        tdist.init_process_group(backend="nccl" if dev.type == "cuda" else "gloo", init_method=f"tcp://127.0.0.1:{port}",
                                 rank=0, world_size=1, **({"device_id": dev} if dev.type == "cuda" else {}))
    return tdist


def run_train(cfg_fn, dev, dist, rank, dtype, steps, warmup, ddp_kwargs=None, probe=None, freeze_dead=False):
    """BASELINE.json configs 3/4: one step = the reference's train loop body for one data batch
    (core/catre/engine/engine.py:293-355): K_ITER x (pose-apply, forward + loss, backward, optimizer step), the fed-back
    pose detached.  With N > 1 the model is wrapped in DistributedDataParallel exactly like
    core/catre/main_catre.py:154-160 and gradients are all-reduced over RCCL (17.19 MB per backward).  `ddp_kwargs` (a dict,
    possibly empty) forces the same wrap on a single rank (`--ddp-world1`: the reducer's cost without the wire); `probe` (a
    dict) receives the gradient order of one extra, untimed backward against the wrapper's bucket plan.
    -> seconds for `steps` steps on this rank (barrier + synchronize on both sides)."""
    from catre_amd import synth
    from catre_amd.batching import batch_updater_test
    from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes

    cfg = cfg_fn(str(dev))
    amp = dtype == "bf16"
    if dtype == "split":
        cfg.MODEL.CATRE.COMPUTE_DTYPE = "split"  # hi + lo bf16 operands, three products: fp32-grade GEMMs on the bf16 pipe
    cfg.SOLVER.OPTIMIZER_CFG = dict(type="Ranger", lr=1e-5, weight_decay=0, clean_grads=True)  # shipped optimiser, fused HIP step
    if freeze_dead:
        # the six `norm.*` tensors no forward uses (conv_out_per_rot_head.py:92, fc_trans_size_head.py:28) taken out of the
        # reducer: with them DDP's find_unused_parameters path waits for its used-parameter bitmap and copies it to the host
        # at the end of every backward (reducer.cpp finalize_bucket_dense) - a device synchronisation per iteration
        cfg.MODEL.CATRE.FREEZE_UNUSED_NORM = True
    model, opt = build_model_optimizer(cfg, is_test=False)
    sd = synth.recipe_state_dict(expected_state_shapes(cfg))
    model.load_state_dict({k: v.to(dev) for k, v in sd.items()}, strict=True)
    model.train()
    net = model
    if dist is not None or ddp_kwargs is not None:
        from torch.nn.parallel import DistributedDataParallel

        net = DistributedDataParallel(model, device_ids=[dev.index], broadcast_buffers=False, find_unused_parameters=True,
                                      **(ddp_kwargs or {}))
    batch = {k: v.to(dev) for k, v in synth.make_inputs(B_PER_GPU, N_PTS, M_PTS, seed=2000 + rank).items()}
    ang = torch.arange(1, 314, dtype=torch.float32) * (2 * 3.141592653589793 / 314)
    sym = torch.zeros(313, 3, 3)
    sym[:, 0, 0], sym[:, 0, 2], sym[:, 1, 1], sym[:, 2, 0], sym[:, 2, 2] = ang.cos(), ang.sin(), 1.0, -ang.sin(), ang.cos()
    sym = sym.numpy()  # 313 y-axis symmetry rotations (MAX_SYM_DISC_STEP=0.01, lib/pysixd/misc.py:220-231)
    sym_info = [sym if (i % 6) in (0, 1, 3) else None for i in range(B_PER_GPU)]  # bottle / bowl / can (ref/nocs.py:138-158)

    gprobe = [None]

    def one_step():
        b = dict(batch)
        poses_est = scales_est = None
        for it in range(1, K_ITER + 1):
            batch_updater_test(cfg, b, poses_est=poses_est, scales_est=scales_est)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):  # engine.py:304 (SOLVER.AMP.ENABLED)
                out, ld = net(b["x"], b["tfd_kps"], init_pose=b["obj_pose_est"], init_scale=b["obj_scale_est"],
                              K_zoom=b["K"], gt_ego_rot=b["gt_rot"], gt_trans=b["gt_trans"], gt_scale=b["gt_scale"],
                              obj_kps=b["obj_kps"], mean_scales=b["obj_mean_scales"], sym_info=sym_info, do_loss=True,
                              cur_iter=it)
            poses_est, scales_est = out[f"pose_{it}"].detach(), out[f"scale_{it}"].detach()
            if gprobe[0] is not None and it == K_ITER:
                gprobe[0].begin()
            sum(ld.values()).backward()
            if gprobe[0] is not None and it == K_ITER:
                gprobe[0].end()
            opt.step()
            opt.zero_grad(set_to_none=True)
        return poses_est

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = one_step()
    barrier()
    dt = time.perf_counter() - t0
    assert torch.isfinite(last).all()
    if probe is not None:  # one more step, untimed, with a hook behind every gradient accumulation
        gprobe[0] = GradOrderProbe(list(model.named_parameters()))
        one_step()
        torch.cuda.synchronize(dev)
        cap = (ddp_kwargs or {}).get("bucket_cap_mb", 25)
        plan = ddp_bucket_plan(model.named_parameters(), cap)
        probe["bucket_cap_mb"] = cap
        probe["gradients_accumulated"] = len(gprobe[0].names)
        probe["first_gradients"] = gprobe[0].names[:3]
        probe["last_gradients"] = gprobe[0].names[-3:]
        probe["buckets_in_launch_order"] = gprobe[0].report(plan)
        gprobe[0].remove()
        gprobe[0] = None
    return dt


def ddp_world1_block(cfg_fn, dev, base_ms_it, steps=3, warmup=1):
    """Scaling evidence a 1-GPU box can produce (BASELINE configs 4/5 wrap every rank's model like this): the reference's
    `DistributedDataParallel(model, device_ids=[local_rank], broadcast_buffers=False, find_unused_parameters=True)`
    (core/catre/main_catre.py:154-160) around the same B=256 training step on a one-rank RCCL group - everything a rank of an
    8-GPU job pays besides the wire: the find_unused_parameters graph walk after every forward, 68 reducer hooks, the copies
    of 17.19 MB of gradients into the buckets, one all-reduce launch per bucket, K times per batch."""
    tdist = init_world1_group(dev)
    out = {"what": "the train_fp32 step with the reference's DDP wrap on a one-rank RCCL group (main_catre.py:154-160); "
                   "overhead = this minus the unwrapped step of the same run", "steps": steps, "warmup": warmup}
    probe = {}
    t = run_train(cfg_fn, dev, None, 0, "fp32", steps, warmup, ddp_kwargs={}, probe=probe)
    ms = t / steps / K_ITER * 1e3
    out["ms_per_iteration"] = round(ms, 3)
    out["overhead_ms_per_iteration"] = round(ms - base_ms_it, 3)
    out["gradient_order_vs_buckets"] = probe
    torch.cuda.empty_cache()
    t = run_train(cfg_fn, dev, None, 0, "fp32", steps, warmup, ddp_kwargs={"gradient_as_bucket_view": True})
    ms_v = t / steps / K_ITER * 1e3
    out["gradient_as_bucket_view"] = {"ms_per_iteration": round(ms_v, 3), "overhead_ms_per_iteration": round(ms_v - base_ms_it, 3)}
    torch.cuda.empty_cache()
    probe4 = {}
    t = run_train(cfg_fn, dev, None, 0, "fp32", steps, warmup, ddp_kwargs={"bucket_cap_mb": 4, "gradient_as_bucket_view": True},
                  probe=probe4)
    ms_4 = t / steps / K_ITER * 1e3
    out["bucket_cap_4MB_bucket_view"] = {"ms_per_iteration": round(ms_4, 3), "overhead_ms_per_iteration": round(ms_4 - base_ms_it, 3),
                                         "gradient_order_vs_buckets": probe4}
    torch.cuda.empty_cache()
    t = run_train(cfg_fn, dev, None, 0, "fp32", steps, warmup, ddp_kwargs={"gradient_as_bucket_view": True}, freeze_dead=True)
    ms_f = t / steps / K_ITER * 1e3
    out["bucket_view_dead_norms_frozen"] = {
        "what": "same wrap, cfg.MODEL.CATRE.FREEZE_UNUSED_NORM=True: the six never-used `norm.*` tensors with requires_grad=False (they stay in the state_dict): every "
                "reducer parameter is then used in every iteration and the find_unused_parameters path never synchronises "
                "the device on its used-parameter bitmap",
        "ms_per_iteration": round(ms_f, 3), "overhead_ms_per_iteration": round(ms_f - base_ms_it, 3)}
    # expected 8-GPU iteration (UNMEASURED: the pool has no multi-GPU box): the wrapped iteration + the part of the ring
    # all-reduce that the rest of the backward cannot hide.  Ring time 2 (N-1)/N x bytes / per-link bandwidth (xGMI, 153 GB/s)
    wire_ms = 2 * 7 / 8 * GRAD_ALLREDUCE_BYTES / 153e9 * 1e3
    out["expected_8gpu"] = {"unmeasured": True, "ring_allreduce_ms": round(wire_ms, 3),
                            "ms_per_iteration_if_fully_exposed": round(ms + wire_ms, 3),
                            "note": "default 25 MiB buckets: the one big bucket closes with the LAST gradient of the backward "
                                    "(see gradient_order_vs_buckets), so its all-reduce is exposed; 4 MiB buckets overlap all "
                                    "but the last one"}
    tdist.barrier()
    return out


def bench_train(args, world, rank, dev, dist, cfg_fn):
    amp = args.dtype == "bf16"
    split = args.dtype == "split"
    ddp1 = None
    if args.ddp_world1 and world == 1:
        init_world1_group(dev)
        ddp1 = {"bucket_cap_mb": args.bucket_cap_mb, "gradient_as_bucket_view": bool(args.bucket_view)}
    probe = {} if ddp1 is not None else None
    dt = run_train(cfg_fn, dev, dist, rank, args.dtype, args.steps, args.warmup, ddp_kwargs=ddp1, probe=probe)
    dt, per_rank_ms, ranks_seen = rank_stats(dist, dev, dt)
    comm = comm_info(dist, dev, world, rank, dev.index)
    if rank == 0:
        value = world * B_PER_GPU * K_ITER * args.steps / dt
        emit(({
            "ranks_seen": ranks_seen, "per_rank_ms": per_rank_ms, "comm": comm, "shared_gpu": SHARE_GPU,
            "allreduce_bytes_per_step": GRAD_ALLREDUCE_BYTES * K_ITER if world > 1 else 0,
            "ddp_world1": dict(ddp1, gradient_order_vs_buckets=probe) if ddp1 is not None else None,
            "ms_per_iteration": round(dt / args.steps / K_ITER * 1e3, 3),
            "metric": "pose-refine TRAIN iters/sec (B=256, N=1024, K=4)" + (" [bf16 autocast]" if amp else " [split-bf16 GEMMs]" if split else ""),
            "value": round(value, 1),
            "unit": "object-iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if amp else ("f32+bf16x3" if split else "f32"), "data": "synthetic",
            "config": {"workload": "B=256 objects/GPU, N=M=1024, K=4 x (forward + loss + backward + fused Ranger step); "
                                   + ("bf16-operand forward / dgrad / wgrad GEMMs under torch.autocast; " if amp else "")
                                   + ("forward / dgrad / wgrad GEMMs as split-bf16 (hi+lo, three products) MFMAs, fp32 results; " if split else "")
                                   +
                                   "half the objects y-symmetric with 313 candidate rotations; "
                                   + ("DDP gradient all-reduce over RCCL" if world > 1 else
                                      ("single rank inside DistributedDataParallel on a one-rank RCCL group" if ddp1 is not None
                                       else "single rank, no DistributedDataParallel wrapper (main_catre.py:154 wraps only when world > 1)")),
                       "objects_per_gpu": B_PER_GPU, "N": N_PTS, "M": M_PTS, "K": K_ITER},
        }))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    elif ddp1 is not None:
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-extra", action="store_true", help="skip the bounded fp32 training measurement of the default line")
    ap.add_argument("--no-small-extra", action="store_true", help="skip the single-image (B=1 / B=4) measurement of the default line")
    ap.add_argument("--mode", choices=("refine", "train"), default="refine",
                    help="refine: the headline inference metric (default); train: configs 3/4 (fwd+loss+bwd+step)")
    ap.add_argument("--dtype", choices=("fp32", "split", "bf16"), default="fp32",
                    help="fp32: fp32 MFMA everywhere (the headline); split: the layers holding 98 %% of the FLOPs as split-bf16 "
                         "(hi+lo, 3 products) MFMAs, same 2e-5 parity; bf16: bf16 operands (BASELINE config 5)")
    ap.add_argument("--shape", choices=("headline", "config5", "config2"), default="headline",
                    help="headline: B=256, N=M=1024, K=4; config5: N=2048 observed, M=1024, K=8; config2: B=64 (BASELINE "
                         "configs[1] as written - the headline metric is quoted at B=256)")
    ap.add_argument("--no-split-extra", action="store_true", help="skip the split-mode measurement of the default line")
    ap.add_argument("--no-ddp-extra", action="store_true", help="skip the DDP-world-1 measurement of the default line")
    ap.add_argument("--ddp-world1", action="store_true",
                    help="--mode train on ONE GPU inside the reference's DistributedDataParallel wrap (one-rank RCCL group)")
    ap.add_argument("--bucket-cap-mb", type=float, default=25, help="--ddp-world1: DDP bucket_cap_mb (torch default 25)")
    ap.add_argument("--bucket-view", action="store_true", help="--ddp-world1: gradient_as_bucket_view=True")
    args = ap.parse_args()
    global N_PTS, K_ITER, B_PER_GPU
    if args.shape == "config5":
        N_PTS, K_ITER = 2048, 8
    if args.shape == "config2":
        B_PER_GPU = 64
    bf16 = args.dtype == "bf16"
    split = args.dtype == "split"
    # split arithmetic spends three bf16 MFMAs per fp32-accurate product: its matrix ceiling is a third of the bf16 peak
    mfma_peak = BF16_MFMA_PEAK_TFLOPS if bf16 else (BF16_MFMA_PEAK_TFLOPS / 3 if split else FP32_MFMA_PEAK_TFLOPS)

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not ("WORLD_SIZE" not in os.environ and args.gpus > 1):
        claim_stdout()   # a rank (or the single process): from here on only `emit` reaches the real stdout
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher (one rank per GPU, like main_catre.py:186-193's
        # detectron2 `launch`), forward the ranks' exit status
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={world} of the launcher")
    if DRYRUN:
        return bench_dryrun(args, world, rank)
    if SHARE_GPU:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if SHARE_GPU:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)  # RCCL on ROCm; used for barrier / max only

    from catre_amd import hip, synth
    from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
    from catre_amd.config import default_cfg

    hip.load()  # fail loudly if the HIP library is missing

    def cfg_fn(device):
        return default_cfg(num_pcl=N_PTS, num_kps=M_PTS, n_iter=K_ITER, device=device)

    if args.mode == "train":
        return bench_train(args, world, rank, dev, dist, cfg_fn)

    cfg = cfg_fn(str(dev))
    model, _ = build_model_optimizer(cfg, is_test=True)
    sd = synth.recipe_state_dict(expected_state_shapes(cfg))
    model.load_state_dict({k: v.to(dev) for k, v in sd.items()}, strict=True)
    model.eval()
    model.cfg.MODEL.CATRE.COMPUTE_DTYPE = args.dtype
    # each rank refines its own shard of the global batch (disjoint seeds)
    batch = {k: v.to(dev) for k, v in synth.make_inputs(B_PER_GPU, N_PTS, M_PTS, seed=1000 + rank).items()}

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        model.refine(batch, n_iter=K_ITER)
    nrec = args.steps * K_ITER
    hip.profile_kernel("trunk", nrec)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = model.refine(batch, n_iter=K_ITER)
    barrier()
    dt = time.perf_counter() - t0
    trunk_ms = hip.profile_collect(nrec)
    hip.profile_kernel(None, 0)
    assert torch.isfinite(out[f"pose_{K_ITER}"]).all()

    # stand-alone channel-wise max-pool [B,1024,N] -> [B,1024] (the HBM-bound figure the north star asks
    # for; in the fused path these bytes never exist).  Outside the timed region, rank 0 only.
    maxpool = None
    if rank == 0:
        from catre_amd.runtime import colmax

        xs = torch.randn(B_PER_GPU, 1024, N_PTS, device=dev)
        for _ in range(3):
            colmax(xs)
        reps = 20
        hip.profile_kernel("colmax", reps)
        for _ in range(reps):
            ymax = colmax(xs)
        ms = hip.profile_collect(reps)
        hip.profile_kernel(None, 0)
        assert torch.equal(ymax, xs.max(2)[0])
        nbytes = 4 * B_PER_GPU * 1024 * N_PTS + 4 * B_PER_GPU * 1024
        avg = sum(ms) / len(ms)
        maxpool = {"kernel": "k_colmax", "bytes": nbytes, "avg_launch_ms": round(avg, 4),
                   "achieved_GBps": round(nbytes / (avg * 1e-3) / 1e9, 1), "peak_GBps": 8000.0,
                   "frac": round(nbytes / (avg * 1e-3) / 8e12, 4)}
        del xs, ymax

    # the same batch through the split compute mode (outside the timed region, rank 0 only): throughput and its
    # deviation from the fp32 kernels' result - reported next to the headline, never as `value`
    split_extra = None
    if rank == 0 and args.dtype == "fp32" and not args.no_split_extra:
        model.cfg.MODEL.CATRE.COMPUTE_DTYPE = "split"
        for _ in range(2):
            osp = model.refine(batch, n_iter=K_ITER)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            osp = model.refine(batch, n_iter=K_ITER)
        torch.cuda.synchronize(dev)
        dts = time.perf_counter() - t1
        model.cfg.MODEL.CATRE.COMPUTE_DTYPE = "fp32"
        dev_max = max(float((osp[f"pose_{K_ITER}"] - out[f"pose_{K_ITER}"]).abs().max()),
                      float((osp[f"scale_{K_ITER}"] - out[f"scale_{K_ITER}"]).abs().max()))
        split_extra = {"what": "same batch, COMPUTE_DTYPE='split': STN conv2/conv3 (+fstn conv1), trunk conv3/conv4 and rot-head layers 0/1 as "
                               "split-bf16 (hi+lo, 3 products) MFMAs with fp32 accumulation, everything else the fp32 "
                               "kernels; the parity tests hold it to the same 2e-5 (contract 1e-4) as the fp32 path",
                       "value": round(B_PER_GPU * K_ITER * args.steps / dts, 1), "unit": "object-iterations/s (1 GPU)",
                       "ms_per_step": round(dts / args.steps * 1e3, 3),
                       "max_abs_diff_vs_fp32_after_K": dev_max}

    # The evaluator's operating point next to the headline (outside the timed region, rank 0 of a 1-GPU run only): one
    # image = a handful of objects per call (catre_evaluator.py:292-311).  Object 0 of the batch alone, K refine
    # iterations, and the check that it gets the very bits it got inside the batch of 256 (latency path).
    small_extra = None
    if rank == 0 and world == 1 and args.dtype == "fp32" and args.shape == "headline" and not args.no_small_extra:
        small_extra = {"what": "K=4 refine of 1 / 4 objects (the reference evaluator's shape: one image per call), fp32 "
                               "kernels, whole K loop as one C call; outside the timed region", "unit": "ms per K=4 refine"}
        for nb in (1, 4):
            sub = {k: v[:nb].contiguous() for k, v in batch.items()}
            for _ in range(10):
                osm = model.refine(sub, n_iter=K_ITER)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(200):
                osm = model.refine(sub, n_iter=K_ITER)
            torch.cuda.synchronize(dev)
            small_extra[f"B{nb}_ms"] = round((time.perf_counter() - t1) / 200 * 1e3, 4)
            small_extra[f"B{nb}_bitwise_equal_to_batch_of_256"] = bool(
                torch.equal(osm[f"pose_{K_ITER}"], out[f"pose_{K_ITER}"][:nb])
                and torch.equal(osm[f"scale_{K_ITER}"], out[f"scale_{K_ITER}"][:nb]))

    # BASELINE.json config 3 next to the headline (outside the timed region, rank 0 of a 1-GPU run only): the training
    # step of engine.py:293-355 at the same batch, fp32, a few steps - so the driver's record carries a train number too
    train_extra = None
    if rank == 0 and world == 1 and args.dtype == "fp32" and args.shape == "headline" and not args.no_train_extra:
        tsteps = 3
        tdt = run_train(cfg_fn, dev, None, 0, "fp32", tsteps, 1)
        train_extra = {"what": "BASELINE config 3: K=4 x (pose-apply, forward + device-side loss, backward, fused Ranger step) of "
                               "the same B=256, N=M=1024 batch, fp32 kernels, half the objects y-symmetric (313 candidates), "
                               "single rank WITHOUT the DistributedDataParallel wrapper (the reference wraps only when world > 1, "
                               "main_catre.py:154; `ddp_world1` below is the same step inside the wrapper); 1 warm-up + 3 timed steps",
                       "value": round(B_PER_GPU * K_ITER * tsteps / tdt, 1), "unit": "training object-iterations/s (1 GPU)",
                       "ms_per_step": round(tdt / tsteps * 1e3, 3), "ms_per_iteration": round(tdt / tsteps / K_ITER * 1e3, 3)}

        flop_it, src = train_flops_per_iteration()
        if flop_it:
            tfs = flop_it / (tdt / tsteps / K_ITER) / 1e12
            train_extra.update({"mfma_gflop_per_iteration": round(flop_it / 1e9, 1), "mfma_gflop_source": src,
                                "tflops": round(tfs, 1), "path_frac_of_mfma_peak": round(tfs / FP32_MFMA_PEAK_TFLOPS, 4)})

        if not args.no_ddp_extra:
            torch.cuda.empty_cache()
            train_extra["ddp_world1"] = ddp_world1_block(cfg_fn, dev, train_extra["ms_per_iteration"])

        # ... and the same step under torch.autocast (engine.py:304, SOLVER.AMP.ENABLED - BASELINE config 5's arithmetic):
        # bf16-operand GEMMs, fp32 accumulation / statistics / SO(3); a bf16-class number, not the fp32 contract
        torch.cuda.empty_cache()  # the fp32 run's 8 GB of saved activations: start from a clean allocator
        adt = run_train(cfg_fn, dev, None, 0, "bf16", tsteps, 2)
        train_extra["autocast_bf16"] = {"value": round(B_PER_GPU * K_ITER * tsteps / adt, 1),
                                        "ms_per_iteration": round(adt / tsteps / K_ITER * 1e3, 3),
                                        "steps": tsteps, "warmup": 2}

    dt, per_rank_ms, ranks_seen = rank_stats(dist, dev, dt)
    comm = comm_info(dist, dev, world, rank, local_rank)

    if rank == 0:
        obj_iters = world * B_PER_GPU * K_ITER * args.steps
        value = obj_iters / dt
        trunk_avg_ms = sum(trunk_ms) / max(len(trunk_ms), 1)
        trunk_flops = B_PER_GPU * (N_PTS + M_PTS) * TRUNK_FLOPS_PER_POINT  # all points of both clouds per launch
        achieved = trunk_flops / (trunk_avg_ms * 1e-3) / 1e12 if trunk_ms else None
        # HBM bytes of the dominant kernel cannot be counted inside this run (PMC passes serialise the kernels and need
        # rocprofv3): the figure is the one collected from this build by profiles/pmc.sh (separate --pmc passes, gfx950
        # FETCH_SIZE x2 correction) and committed next to its summary; `traffic_source` says so in the line itself
        traffic, traffic_source = None, None
        import glob
        pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_trunk_hbm_bytes.json")))  # newest round's pass
        if pmcs and args.dtype == "fp32" and args.shape == "headline":
            pmc = pmcs[-1]
            tag = os.path.basename(pmc)[:3]
            with open(pmc) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch")
            traffic_source = (f"profiles/{os.path.basename(pmc)}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                              f"build (profiles/pmc.sh, profiles/{tag}_pmc_summary.csv), not measured in this run; algorithmic "
                              f"bytes per launch = {B_PER_GPU * (N_PTS + M_PTS) * 12}")
        path_flops = flops_per_object_iteration(N_PTS, M_PTS)
        line = {
            "metric": f"pose-refine iters/sec (B={B_PER_GPU}, N={N_PTS}, K={K_ITER})"
                      + (" [bf16 operands]" if bf16 else (" [split-bf16 GEMMs]" if split else "")),
            "value": round(value, 1),
            "unit": "object-iterations/s",
            "n_gpus": world,
            "ranks_seen": ranks_seen,  # ranks that reported through the job's backend (RCCL for N > 1)
            "per_rank_ms": per_rank_ms,
            "comm": comm,
            "shared_gpu": SHARE_GPU,  # true only under CATRE_BENCH_SHARE_GPU=1 (all ranks on GPU 0: plumbing test, not a scaling run)
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16" if bf16 else ("f32+bf16x3" if split else "f32"),
            "data": "synthetic",
            "config": {
                "workload": f"B={B_PER_GPU} objects/GPU, N={N_PTS} observed + M={M_PTS} prior points, K={K_ITER} refine iterations, "
                            "forward-only (eval loop of catre_evaluator.py:292-311), "
                            + ("bf16 MFMA operands / fp32 accumulate" if bf16 else
                               ("fp32 results; STN conv2/conv3 (+fstn conv1), trunk conv3/conv4 and rot-head layers 0/1 as split-bf16 (hi+lo, "
                                "three products) MFMAs, the rest fp32 MFMA" if split else "fp32 MFMA")),
                "objects_per_gpu": B_PER_GPU, "N": N_PTS, "M": M_PTS, "K": K_ITER,
                "parallelism": f"batch-sharded x{world} (no data-path collective)",
            },
            "path_tflops": round(value * path_flops / 1e12, 2),
            "path_frac_of_mfma_peak": round(value * path_flops / 1e12 / (mfma_peak * world), 4),
            "roofline": {
                "kernel": "k_trunk_bf2" if bf16 else ("k_trunk_split" if split else "k_trunk4"),  # (full grids: the forms that fill the chip)
                "bound": "mfma",
                "achieved": round(achieved, 2) if achieved else None,
                "peak": mfma_peak,
                "unit": "TFLOP/s",
                "frac": round(achieved / mfma_peak, 4) if achieved else None,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "avg_launch_ms": round(trunk_avg_ms, 4),
                "launches_timed": len(trunk_ms),
                "flops_per_launch": trunk_flops,
            },
        }
        line["maxpool_standalone"] = maxpool
        if split_extra is not None:
            line["split_mode"] = split_extra
        if small_extra is not None:
            line["single_image"] = small_extra
        if train_extra is not None:
            line["train_fp32"] = train_extra
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg_fn, sd)
        emit(line)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    elif torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()  # the one-rank group of the ddp_world1 block


if __name__ == "__main__":
    main()
