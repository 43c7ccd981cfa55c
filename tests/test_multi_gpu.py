"""N > 1 on hardware (the reference's only parallelism is DDP, core/catre/main_catre.py:154-160).

Every test here SKIPS on a box with fewer than two GPUs - the pool hands out one GPU per box, so on a normal run nothing
changes - and produces scaling-readiness evidence by itself on any multi-GPU lease: RCCL gradient means, the `bench.py`
launcher with real ranks, the all-reduce payload of the train line."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NGPU = torch.cuda.device_count() if torch.cuda.is_available() else 0
need2 = pytest.mark.skipif(NGPU < 2, reason=f"needs >= 2 GPUs on the box, found {NGPU}")


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("CATRE_BENCH_DRYRUN", None)
    return env


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _ddp(tmp_path, share):
    out = tmp_path / "ddp.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "ddp_worker.py"), str(out)]
    env = _env()
    env["CATRE_SHARE_GPU"] = "1" if share else "0"
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    v = json.loads(out.read_text())
    assert v["world"] == 2 and v["tensors"] == 68 and v["elements"] == 4297175 - (1024 + 1024 - 256 - 128) * 2
    assert v["ranks_differ_by"] > 0, "both ranks computed the same gradients: the batches were not distinct"
    assert v["worst_rel_to_max"] <= 1e-6, v
    return v


@need2
def test_ddp_over_rccl_averages_the_single_gpu_gradients(tmp_path):
    """2 ranks, different batches: the gradients DDP leaves on every rank == mean of the two ranks' single-GPU gradients
    (<= 1e-6 of each tensor's max: the all-reduce sums two fp32 numbers and divides by 2)."""
    assert _ddp(tmp_path, share=False)["backend"] == "nccl"


def test_ddp_two_processes_sharing_one_gpu_average_their_gradients(tmp_path):
    """The same check on ANY GPU box: two rank processes drive GPU 0 concurrently and exchange gradients through gloo (RCCL
    refuses two ranks on one device).  Everything but the transport is the real thing: torch.distributed.run, two HIP
    contexts, `DistributedDataParallel(find_unused_parameters=True)` hooks firing on HIP-computed gradients of 68 tensors
    while 6 stay unused, bucketed all-reduce, optimizer step on the wrapped module (main_catre.py:154-160)."""
    assert _ddp(tmp_path, share=True)["backend"] == "gloo"


def _bench(*args, env=None):
    e = _env()
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, env=e, capture_output=True,
                       text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@need2
def test_bench_refine_with_two_real_ranks():
    line = _bench("--gpus", "2", "--steps", "3", "--warmup", "1")
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and len(line["per_rank_ms"]) == 2
    assert line["scaling"] == "weak" and line["value"] > 0
    assert line["comm"]["backend"] == "nccl" and len(line["comm"]["ranks"]) == 2
    assert {r["local_rank"] for r in line["comm"]["ranks"]} == {0, 1}


@need2
def test_bench_train_with_two_real_ranks_reports_the_allreduce_payload():
    line = _bench("--gpus", "2", "--mode", "train", "--steps", "2", "--warmup", "1")
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2
    assert line["allreduce_bytes_per_step"] == 4 * 4297175 * 4   # K=4 backward passes x 17.19 MB of fp32 gradients
    assert line["value"] > 0 and line["comm"]["backend"] == "nccl"


def test_bench_launcher_with_two_rank_processes_sharing_one_gpu():
    """`python bench.py --gpus 2` end to end on a 1-GPU box (CATRE_BENCH_SHARE_GPU=1: both ranks on GPU 0, gloo): the
    self-launch, rendezvous, per-rank batches, barrier + synchronize bracketing, MAX-over-ranks timing and rank-0 JSON with
    REAL refine / training work on the device.  The line is marked `shared_gpu` - it is plumbing evidence, never a scaling
    number."""
    env = {"CATRE_BENCH_SHARE_GPU": "1"}
    line = _bench("--gpus", "2", "--steps", "2", "--warmup", "1", env=env)
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and len(line["per_rank_ms"]) == 2 and line["shared_gpu"] is True
    assert line["comm"]["backend"] == "gloo" and {r["rank"] for r in line["comm"]["ranks"]} == {0, 1}
    assert line["value"] > 0 and line["roofline"]["frac"] > 0
    line = _bench("--gpus", "2", "--mode", "train", "--steps", "1", "--warmup", "1", env=env)
    assert line["ranks_seen"] == 2 and line["shared_gpu"] is True and line["allreduce_bytes_per_step"] == 4 * 4297175 * 4
    assert line["value"] > 0


def test_single_gpu_line_carries_the_comm_block():
    """On any GPU box: the default line names the collective library version and the rank -> device map (world 1)."""
    line = _bench("--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-train-extra", "--no-small-extra")
    c = line["comm"]
    assert line["ranks_seen"] == 1 and len(c["ranks"]) == 1 and c["ranks"][0]["rank"] == 0
    assert "rccl_version" in c and "gfx950" in c["ranks"][0]["gcn_arch"]
