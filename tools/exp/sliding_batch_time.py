"""Several queries of one length in one pass over the ragged corpus (k_sliding.hip, round 5): time of a batch of 1, 2, 4, 8
queries through LBAudioDetectiveCorpusQueryBatchKeysDevice against eight single scans; results checked against each other."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import lbaudiodetective_amd as lb
from oracle import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--nq", type=int, nargs="+", default=[21, 5])
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--json")
a = ap.parse_args()
SEED = 0x4C424145
counts = O.synth_ragged_counts(SEED, 0, a.n, 20, 70)
total = int(counts.sum())
packed = lb.synth_ragged_corpus_device(SEED, 0, counts, 200)
c = lb.Corpus.ragged(200, a.n, total)
c.append_ragged_packed_device(packed, counts)
del packed
c.set_bound_pruning(False)
out = []
for nq in a.nq:
    fps = []
    for k in range(8):
        e = min(100_000 * (k + 1) + 777, a.n - 1)
        src = O.synth_entry(SEED, e, max(int(counts[e]), nq), 200)
        fps.append(lb.Fingerprint.from_bools(src[:nq]))
    keys = torch.zeros(8, dtype=torch.int64, device="cuda")
    single = torch.zeros(8, dtype=torch.int64, device="cuda")
    for i, f in enumerate(fps):
        c.query_key_device(f, single[i:i + 1])
    torch.cuda.synchronize()
    row = {"n_entries": a.n, "records": total, "n_query": nq}
    for b in (1, 2, 4, 8):
        for _ in range(2):
            c.query_batch_keys_device(fps[:b], keys)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            c.query_batch_keys_device(fps[:b], keys)
        e1.record()
        torch.cuda.synchronize()
        row[f"batch_of_{b}_ms"] = round(e0.elapsed_time(e1) / a.reps, 4)
        assert torch.equal(keys[:b], single[:b]), (nq, b, keys.tolist(), single.tolist())
    row["eight_over_one"] = round(row["batch_of_8_ms"] / row["batch_of_1_ms"], 2)
    row["GBps_algorithmic_per_query_in_batch_of_8"] = round(25 * total * 8 / row["batch_of_8_ms"] / 1e6, 1)
    print(json.dumps(row))
    out.append(row)
if a.json:
    json.dump(out, open(a.json, "w"), indent=1)
