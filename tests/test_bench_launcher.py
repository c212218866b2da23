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


def test_world_size_eight_with_planted_index_off_rank_zero():
    """BASELINE configs[3]'s shard arithmetic at the size the driver will run it: 10 M entries over EIGHT ranks
    (1.25 M each), the planted match at 7 777 777 -- rank 6, never rank 0 -- found through the MAX reduction of the
    (score, ~index) keys."""
    out = _run("--gpus", "8", "--backend", "gloo", "--clips", "0")
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert r["n_gpus"] == 8 and r["rccl_ranks"] == 8 and sorted(d["rank"] for d in r["devices"]) == list(range(8))
    c = r["compare"]
    assert c["entries_per_rank"] == [1_250_000] * 8 and c["planted_rank"] == 6
    assert c["best_index"] == c["planted_index"] == 7_777_777 and abs(c["best_score"] - 0.93) < 1e-6
    assert r["self_check"]["ok"]


def test_world_size_eight_with_empty_shards():
    """Fewer entries than ranks: three ranks own one entry, five own none and must still join the reduction with
    the neutral key; the planted entry (index 7 777 777 % 3 = 1) wins."""
    out = _run("--gpus", "8", "--backend", "gloo", "--clips", "0", "--corpus", "3")
    assert out.returncode == 0, out.stderr[-2000:]
    c = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])["compare"]
    assert sorted(c["entries_per_rank"]) == [0] * 5 + [1] * 3 and sum(c["entries_per_rank"]) == 3
    assert c["best_index"] == c["planted_index"] == 1 and abs(c["best_score"] - 0.93) < 1e-6


def test_force_dist_takes_the_collective_branches_at_one_rank():
    """`--force-dist` at world size 1 (gloo here; tests/test_gpu_ragged.py::test_bench_collective_paths_on_nccl_at_one_rank runs the same switch on the nccl backend):
    the process group exists, the reduction goes through torch.distributed and the line says so."""
    out = _run("--gpus", "1", "--force-dist", "--backend", "gloo", "--clips", "0", "--corpus", "2000")
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert r["n_gpus"] == 1 and r["rccl_ranks"] == 1 and "force_dist" in r and [d["rank"] for d in r["devices"]] == [0]
    c = r["compare"]
    assert c["entries_per_rank"] == [2000] and c["best_index"] == c["planted_index"] and c["allreduce_ms"] > 0


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_stage_times_of_a_chunked_call_are_per_call():
    """Round 3's side legs divided the summed stage times by the number of LAUNCHES and multiplied by the whole batch
    (4 x too fast for a call that runs as four chunks).  Three calls of four chunks, 2 ms + 0.5 ms per chunk: a call
    costs 8 ms + 2 ms, a launch 2 ms + 0.5 ms."""
    b = _bench_module()
    (ms1, ms2, lpc), (k1, k2) = b.per_call_stage_times(3 * 4 * 2.0, 3 * 4 * 0.5, 12, 3)
    assert (ms1, ms2, lpc) == (8.0, 2.0, 4) and (k1, k2) == (2.0, 0.5)
    import pytest
    with pytest.raises(ValueError):
        b.per_call_stage_times(1.0, 1.0, 13, 3)


def test_self_check_rejects_impossible_figures():
    b = _bench_module()
    ok = {"roofline": {"achieved": 1100.0, "peak": 8000.0, "frac": 0.1375, "fp32_canonical_tflops": 99.5,
                       "fp32_peak_tflops": 157.3}, "other_configs": {"x": {"stage1_hbm_frac": 0.03}}}
    assert b.self_check(ok) == []
    bad = {"other_configs": {"configs0_settings": {"stage1_fp32_canonical_tflops": 179.2, "stage1_hbm_frac": 0.1234}},
           "compare_hbm": {"roofline": {"achieved": 9000.0, "peak": 8000.0, "frac": 1.125}},
           "x": [{"layout_GBps": 12000.0}]}
    v = b.self_check(bad)
    assert len(v) == 4 and any("179.2" in m for m in v) and any("achieved 9000.0" in m for m in v)
    assert any("frac = 1.125" in m for m in v) and any("layout_GBps" in m for m in v)
