"""`python bench.py --gpus N` launches its own ranks: the driver calls exactly that form.  Runs the
launcher, the rendezvous, the barrier / max-over-ranks timing and the rank-0 JSON line on CPU (gloo) with
CATRE_BENCH_DRYRUN=1 - no GPU work, no throughput claimed."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = dict(os.environ, CATRE_BENCH_DRYRUN="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True, env=env,
                       timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, lines


@pytest.mark.parametrize("mode", ["refine", "train"])
def test_plain_invocation_spawns_one_rank_per_gpu(mode):
    r, lines = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--mode", mode])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout  # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and len(line["per_rank_ms"]) == 2
    assert line["dryrun"] is True and line["value"] is None
    assert line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    if mode == "train":
        assert line["allreduce_bytes_per_step"] == 4 * 4297175 * 4  # K x 17.19 MB


def test_single_rank_does_not_spawn():
    r, lines = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1


def test_world_size_mismatch_is_an_error():
    r, _ = _run(["--gpus", "2"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "does not match" in r.stderr


def test_ddp_bucket_plan_and_gradient_order_probe_on_a_toy_model():
    """The `ddp_world1` block of the default line relates the order in which gradients are accumulated to the buckets
    `DistributedDataParallel(find_unused_parameters=True)` builds (main_catre.py:154-160).  Here: the plan against the wrapper's
    own reducer on a one-rank gloo group (CPU), and the probe's bookkeeping on a real backward."""
    import torch
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel

    import bench

    torch.manual_seed(0)
    # definition order big -> small -> unused: [1 MiB, cap] over that order, reversed
    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.add_module("0", torch.nn.Linear(600, 600))
            self.add_module("1", torch.nn.Linear(600, 64))
            self.add_module("2", torch.nn.Linear(64, 8))
            self.dead = torch.nn.Linear(4, 4)   # like the 6 unused `norm` tensors of the reference heads

        def forward(self, x):
            return getattr(self, "2")(getattr(self, "1")(getattr(self, "0")(x)))

    model = Toy()
    tdist = bench.init_world1_group(torch.device("cpu"))
    try:
        plan = bench.ddp_bucket_plan(model.named_parameters(), bucket_cap_mb=25)
        assert sorted(k for b in plan for k in b) == sorted(k for k, _ in model.named_parameters())
        # first bucket of the assignment is capped at 1 MiB: 0.weight (1.44 MB) closes it; it is launched LAST
        assert plan[-1][0] == "0.weight" and "dead.weight" in plan[0]
        ddp = DistributedDataParallel(model, broadcast_buffers=False, find_unused_parameters=True)
        probe = bench.GradOrderProbe(list(model.named_parameters()), use_events=False)
        probe.begin()
        ddp(torch.randn(5, 600)).sum().backward()
        probe.end()
        assert len(probe.names) == 6 and probe.names[0].startswith("2.") and probe.names[-1].startswith("0.")
        rep = probe.report(plan)
        assert [r["bucket"] for r in rep] == list(range(len(plan)))
        assert rep[-1]["closes_at_hook"] == 6 and rep[-1]["of_hooks"] == 6      # the first layer's gradients come last
        assert rep[0]["tensors_with_grad"] == len(plan[0]) - 2                  # the dead layer never fires
        probe.remove()
        # the reducer's own bucket count agrees with the plan
        assert len(ddp._get_ddp_logging_data()["bucket_sizes"].split(",")) == len(plan) if hasattr(ddp, "_get_ddp_logging_data") else True
    finally:
        tdist.destroy_process_group()
