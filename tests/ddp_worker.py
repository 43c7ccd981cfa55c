"""One rank of the 2-GPU DDP check (launched by tests/test_multi_gpu.py under torch.distributed.run, RCCL backend).

Every rank builds the same model (seeded weight recipe), takes its OWN batch, and computes the gradients of one training
iteration twice: on the bare module (its single-GPU gradients) and through `DistributedDataParallel(model,
find_unused_parameters=True, broadcast_buffers=False)` - the wrap of core/catre/main_catre.py:154-160.  DDP must hand every
rank the MEAN over ranks of the single-GPU gradients; rank 0 writes the verdict as JSON to argv[1]."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    out_path = sys.argv[1]
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    # CATRE_SHARE_GPU=1: every rank drives GPU 0 and the collectives go through gloo (RCCL refuses two ranks on one
    # device) - the 1-GPU box's stand-in for a 2-GPU node: real processes, real DDP hooks on HIP-computed gradients
    share = os.environ.get("CATRE_SHARE_GPU", "0") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if share:
        dist.init_process_group(backend="gloo")
    else:
        dist.init_process_group(backend="nccl", device_id=dev)

    from catre_amd import synth
    from catre_amd.batching import batch_updater_test
    from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
    from catre_amd.config import default_cfg
    from torch.nn.parallel import DistributedDataParallel

    B, N, M = 6, 256, 128
    cfg = default_cfg(num_pcl=N, num_kps=M, device=str(dev))
    model, opt = build_model_optimizer(cfg, is_test=False)
    sd = synth.recipe_state_dict(expected_state_shapes(cfg))
    model.load_state_dict({k: v.to(dev) for k, v in sd.items()}, strict=True)
    model.train()
    b = {k: v.to(dev) for k, v in synth.make_inputs(B, N, M, seed=900 + rank).items()}
    batch_updater_test(cfg, b)
    sym = [None] * B

    def iteration(net):
        out, ld = net(b["x"], b["tfd_kps"], init_pose=b["obj_pose_est"], init_scale=b["obj_scale_est"], K_zoom=b["K"],
                      gt_ego_rot=b["gt_rot"], gt_trans=b["gt_trans"], gt_scale=b["gt_scale"], obj_kps=b["obj_kps"],
                      mean_scales=b["obj_mean_scales"], sym_info=sym, do_loss=True, cur_iter=1)
        model.zero_grad(set_to_none=True)
        sum(ld.values()).backward()
        return {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}

    solo = iteration(model)
    names = sorted(solo)
    flat = torch.cat([solo[k].reshape(-1) for k in names])
    src = flat.cpu() if share else flat
    gathered = [torch.zeros_like(src) for _ in range(world)]
    dist.all_gather(gathered, src)
    gathered = [g.to(dev) for g in gathered]
    mean = torch.stack(gathered).double().mean(0)
    differ = float((gathered[0] - gathered[-1]).abs().max())   # ranks saw different data

    ddp = DistributedDataParallel(model, device_ids=[local], broadcast_buffers=False, find_unused_parameters=True)
    assert sum(1 for _ in ddp.parameters()) == 74
    got = iteration(ddp)
    assert sorted(got) == names, "DDP changed which parameters receive gradients"
    gflat = torch.cat([got[k].reshape(-1) for k in names]).double()
    worst, off = ("", 0.0), 0
    for k in names:
        n = solo[k].numel()
        ref = mean[off:off + n]
        err = float((gflat[off:off + n] - ref).abs().max()) / (float(ref.abs().max()) + 1e-30)
        if err > worst[1]:
            worst = (k, err)
        off += n
    opt.step()   # the wrapped module steps
    verdict = torch.tensor([worst[1]], device="cpu" if share else dev, dtype=torch.float64)
    dist.all_reduce(verdict, op=dist.ReduceOp.MAX)
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump({"world": world, "tensors": len(names), "elements": int(flat.numel()), "worst_tensor": worst[0],
                       "worst_rel_to_max": float(verdict.item()), "ranks_differ_by": differ,
                       "backend": dist.get_backend()}, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
