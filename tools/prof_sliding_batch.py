#!/usr/bin/env python3
"""Driver for rocprofv3 passes over a BATCH of queries against the ragged corpus (k_sliding.hip, round 5):
    python3 tools/prof_sliding_batch.py [n_entries] [n_query] [batch] [reps]
`batch` queries of n_query sub-fingerprints through LBAudioDetectiveCorpusQueryBatchKeysDevice (four per launch of the task
scan, eight of the systolic scan of short queries)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lbaudiodetective_amd as lb
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 21
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 8
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
SEED = 0x4C424145
counts = O.synth_ragged_counts(SEED, 0, n, 20, 70)
total = int(counts.sum())
packed = lb.synth_ragged_corpus_device(SEED, 0, counts, 200)
c = lb.Corpus.ragged(200, n, total)
c.append_ragged_packed_device(packed, counts)
c.set_bound_pruning(False)
fps = []
for k in range(batch):
    e = min(100_000 * (k + 1) + 777, n - 1)
    fps.append(lb.Fingerprint.from_bools(O.synth_entry(SEED, e, max(int(counts[e]), nq), 200)[:nq]))
keys = torch.zeros(batch, dtype=torch.int64, device="cuda")
for _ in range(reps):
    c.query_batch_keys_device(fps, keys)
torch.cuda.synchronize()
print([lb.Corpus.decode_key(int(k) & (2**64 - 1)) for k in keys.tolist()])
