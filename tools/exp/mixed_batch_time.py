"""Ten queries of ten DIFFERENT lengths in one LBAudioDetectiveCorpusQueryBatchKeysDevice call -- upstream's ten originals
(LBAudioDetectiveTests.m:57-91) -- against 1 M ragged entries, repeated: every length needs its own plan of the task scan
(round 6: a cache of four plans per corpus; round 5 rebuilt the one plan, and drained the GPU, at every change of length)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import lbaudiodetective_amd as lb
from oracle import oracle as O
SEED = 0x4C424145
n = 1_000_000
counts = O.synth_ragged_counts(SEED, 0, n, 20, 70)
total = int(counts.sum())
packed = lb.synth_ragged_corpus_device(SEED, 0, counts, 200)
c = lb.Corpus.ragged(200, n, total)
c.append_ragged_packed_device(packed, counts)
del packed
c.set_bound_pruning(False)
for lengths in ([21, 30, 21, 30, 21, 30, 21, 30, 21, 30], [14, 18, 21, 25, 30, 21, 14, 25, 18, 30], [14, 17, 21, 25, 30, 36, 41, 48, 55, 62]):
    fps = []
    for k, nq in enumerate(lengths):
        e = 90_000 * (k + 1) + 77
        fps.append(lb.Fingerprint.from_bools(O.synth_entry(SEED, e, max(int(counts[e]), nq), 200)[:nq]))
    keys = torch.zeros(len(fps), dtype=torch.int64, device="cuda")
    single = torch.zeros(len(fps), dtype=torch.int64, device="cuda")
    for i, f in enumerate(fps):
        c.query_key_device(f, single[i:i + 1])
    for _ in range(3):
        c.query_batch_keys_device(fps, keys)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        c.query_batch_keys_device(fps, keys)
    e1.record()
    torch.cuda.synchronize()
    assert torch.equal(keys, single)
    print(json.dumps({"lengths": lengths, "distinct": len(set(lengths)), "one_call_ms": round(e0.elapsed_time(e1) / 10, 4)}))
