"""world_size-2 gloo test of the multi-GPU exchange step (no GPU involved): every rank scans
its contiguous shard -- here with the CPU oracle standing in for the HIP scan -- packs
(score, ~index) into one int64 and MAX-all-reduces it through the product's reduction code."""
import os
import struct

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lbaudiodetective_amd import sharded


def test_shard_range_partitions():
    for n in (0, 1, 7, 8, 9, 1000003):
        for ws in (1, 2, 3, 8):
            spans = [sharded.shard_range(n, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, n_entries, planted, q, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    import lbaudiodetective_amd as lb
    # the query exists on rank 0 only; the broadcast hands every rank an identical fingerprint
    fq = lb.broadcast_fingerprint(lb.Fingerprint.from_bools(q) if rank == 0 else None, src=0)
    assert np.array_equal(fq.to_bools(), q)
    begin, end = sharded.shard_range(n_entries, rank, world)
    corpus = O.synth_corpus(77, begin, end - begin, 5, 200)
    for g in planted:                       # identical entries in different shards: lowest index must win
        if begin <= g < end:
            corpus[g - begin] = q
    idx, score = O.corpus_best(q, corpus, 200)
    bits = struct.unpack("<I", struct.pack("<f", score))[0]
    key = torch.tensor([sharded.make_key(bits, begin + idx) if idx >= 0 else 0], dtype=torch.int64)
    sharded.allreduce_best(key)
    ret[rank] = sharded.decode_key(int(key.item()))
    # optional gather of packed sub-fingerprints: ragged per-rank counts, rank order preserved
    mine = torch.full((3 + rank, 5, 32), rank + 1, dtype=torch.uint8)
    allp = sharded.gather_packed(mine)
    assert allp.shape == (sum(3 + r for r in range(world)), 5, 32)
    off = 0
    for r in range(world):
        assert bool((allp[off:off + 3 + r] == r + 1).all())
        off += 3 + r
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("planted", [[1500], [300, 1700], []])
def test_allreduce_top1_two_ranks(oracle, planted):
    n_entries, world = 2000, 2
    q = oracle.synth_entry(5, 123456, 5, 200)
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() + len(planted)) % 2000
    mp.spawn(_worker, args=(world, port, n_entries, planted, q, ret), nprocs=world, join=True)
    corpus = oracle.synth_corpus(77, 0, n_entries, 5, 200)
    for g in planted:
        corpus[g] = q
    want = oracle.corpus_best(q, corpus, 200)
    assert ret[0] == ret[1] == want
    if planted:
        assert want == (min(planted), 1.0)


def _lone_worker(rank, world, port, q, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lbaudiodetective_amd as lb
    calls = []
    real = dist.all_reduce

    def counting(*a, **k):
        calls.append("all_reduce")
        return real(*a, **k)

    dist.all_reduce = counting
    key = torch.tensor([41], dtype=torch.int64)
    sharded.FORCE_COLLECTIVES = False
    sharded.allreduce_best(key)                       # a lone rank skips the collective ...
    skipped = list(calls)
    sharded.FORCE_COLLECTIVES = True                  # ... unless told to take every collective branch (bench.py --force-dist)
    sharded.allreduce_best(key)
    fq = lb.broadcast_fingerprint(lb.Fingerprint.from_bools(q), src=0)
    mine = torch.full((4, 5, 32), 7, dtype=torch.uint8)
    allp = sharded.gather_packed(mine)
    ret[0] = (skipped, list(calls), int(key.item()), bool(np.array_equal(fq.to_bools(), q)), tuple(allp.shape), bool((allp == 7).all()))
    dist.all_reduce = real
    sharded.FORCE_COLLECTIVES = False
    dist.destroy_process_group()


def test_forced_collectives_at_one_rank(oracle):
    """sharded.FORCE_COLLECTIVES (bench.py --force-dist): in a group of ONE rank the helpers really call the backend -- the
    N > 1 branches executed without a second rank -- and leave their data unchanged."""
    q = oracle.synth_entry(5, 99, 5, 200)
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 31500 + os.getpid() % 2000
    mp.spawn(_lone_worker, args=(1, port, q, ret), nprocs=1, join=True)
    skipped, calls, key, same_fp, shape, same_packed = ret[0]
    assert skipped == [] and calls == ["all_reduce"] and key == 41 and same_fp and shape == (4, 5, 32) and same_packed
