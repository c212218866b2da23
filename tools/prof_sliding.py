#!/usr/bin/env python3
"""Driver for rocprofv3 passes over the ragged-corpus scan (k_sliding.hip):
    python3 tools/prof_sliding.py [n_entries] [n_query] [lo] [hi] [reps]
1 M synthetic entries of lo..hi sub-fingerprints, a query of n_query cut out of entry 777 777; `reps` key-only scans.
A device-to-device copy of the record buffer's size follows (known byte count: calibrates FETCH_SIZE / WRITE_SIZE)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lbaudiodetective_amd as lb
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 21
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 20
hi = int(sys.argv[4]) if len(sys.argv) > 4 else 70
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
SEED = 0x4C424145
counts = O.synth_ragged_counts(SEED, 0, n, lo, hi)
total = int(counts.sum())
packed = lb.synth_ragged_corpus_device(SEED, 0, counts, 200)
c = lb.Corpus.ragged(200, n, total)
c.append_ragged_packed_device(packed, counts)
c.set_bound_pruning(False)          # the profile is of the full scan: every sliding offset of every entry
planted = min(777_777, n - 1)
src = O.synth_entry(SEED, planted, max(int(counts[planted]), nq), 200)
q = lb.Fingerprint.from_bools(src[:nq])
key = torch.zeros(1, dtype=torch.int64, device="cuda")
for _ in range(reps):
    c.query_key_device(q, key)
torch.cuda.synchronize()
dst = torch.empty_like(packed)
for _ in range(2):
    dst.copy_(packed)
torch.cuda.synchronize()
print("done", n, nq, "records", total, "record bytes", 32 * total, "copy bytes", packed.numel(), "best", lb.Corpus.decode_key(int(key.item()) & (2**64 - 1)))
