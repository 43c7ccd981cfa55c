"""N>1 path on CPU: two processes over gloo shard a batch, refine their halves (the oracle stands in for
the device compute here - tests may use it), all-gather, and must reproduce the unsharded result."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from catre_amd.sharding import shard_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_and_balance():
    for total in (1, 2, 5, 16, 255, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from catre_amd import synth
    from catre_amd.CATRE_disR_shared import expected_state_shapes
    from catre_amd.config import default_cfg
    from catre_amd.sharding import refine_sharded
    from oracle import catre_oracle as O

    N, M, K, B = 64, 40, 2, 5  # ragged shards: 3 + 2 objects
    cfg = default_cfg(num_pcl=N, num_kps=M, n_iter=K, device="cpu")
    sd = synth.recipe_state_dict(expected_state_shapes(cfg))
    batch = synth.make_inputs(B, N, M, seed=9)

    def refine_fn(local, n_iter):
        with torch.no_grad():
            o = O.refine_k(local, sd, cfg, n_iter=n_iter)
        return {k: v for k, v in o.items() if k.startswith(("pose_", "scale_"))}

    full = refine_sharded(refine_fn, batch, K)
    if rank == 0:
        want = refine_fn(batch, K)
        err = max((full[k] - want[k]).abs().max().item() for k in want)
        q.put((err, {k: tuple(v.shape) for k, v in full.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_refine_matches_unsharded():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    err, shapes = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert shapes["pose_2"] == (5, 3, 4) and shapes["scale_0"] == (5, 3)
    assert err < 1e-6, err
