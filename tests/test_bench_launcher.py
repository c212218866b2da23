"""bench.py's own multi-rank launcher, driven on CPU: `python bench.py --gpus 2 --backend gloo --clips 0`
must start two fresh ranks through torch.distributed.run, rendezvous on 127.0.0.1, reduce the per-shard
(score, ~index) keys with MAX and print ONE JSON line that reports two ranks and the planted index."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *extra], capture_output=True, text=True,
                          env=env, timeout=600)


def test_gpus_flag_launches_that_many_ranks():
    out = _run("--gpus", "2", "--backend", "gloo", "--clips", "0", "--corpus", "2000")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["rccl_ranks"] == 2 and r["backend"] == "gloo" and r["value"] is None
    assert [d["rank"] for d in r["devices"]] == [0, 1] and len({d["pid"] for d in r["devices"]}) == 2
    c = r["compare"]
    assert c["entries_per_rank"] == [1000, 1000]
    assert c["planted_rank"] == 1 and c["best_index"] == c["planted_index"] == 7_777_777 % 2000
    assert abs(c["best_score"] - 0.93) < 1e-6 and c["allreduce_ms"] > 0


def test_world_size_mismatch_is_an_error():
    out = _run("--gpus", "2", "--backend", "gloo", "--clips", "0", env_extra={"WORLD_SIZE": "1", "RANK": "0"})
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr


def test_failed_rank_fails_the_launcher():
    # a dry run that asks for the fingerprint leg is refused by every rank: the parent must report failure
    out = _run("--gpus", "2", "--backend", "gloo", "--clips", "10")
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
