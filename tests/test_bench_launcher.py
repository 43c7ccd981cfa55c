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
