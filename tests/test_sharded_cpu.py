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
