"""Scan time of the ragged corpus (k_sliding.hip): 1 M synthetic entries of 20..70 sub-fingerprints, query of
--nq sub-fingerprints; HIP events on the launch stream around `reps` key-only scans."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import lbaudiodetective_amd as lb
from lbaudiodetective_amd import _native as _N
if os.environ.get("LBAD_LIB"):
    _N.LIB_PATH = os.path.abspath(os.environ["LBAD_LIB"])
from oracle import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--lo", type=int, default=20)
ap.add_argument("--hi", type=int, default=70)
ap.add_argument("--nq", type=int, nargs="+", default=[21])
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--json")
ap.add_argument("--prune", type=int, default=0, help="1: leave the bound pruning of top-1 scans on (default here: off = every offset is evaluated)")
ap.add_argument("--no-match", action="store_true", help="a query that matches nothing in the corpus")
ap.add_argument("--variant", type=int, default=0, help="CorpusSetKernelVariant: 3 forces the split of short entries off to the systolic scan, 4 forbids it")
a = ap.parse_args()
SEED = 0x4C424145
counts = O.synth_ragged_counts(SEED, 0, a.n, a.lo, a.hi)
total = int(counts.sum())
packed = lb.synth_ragged_corpus_device(SEED, 0, counts, 200)
c = lb.Corpus.ragged(200, a.n, total)
c.append_ragged_packed_device(packed, counts)
del packed
c.set_bound_pruning(bool(a.prune))
if a.variant:
    c.set_kernel_variant(a.variant)
key = torch.zeros(1, dtype=torch.int64, device="cuda")
out = []
for nq in a.nq:
    planted = min(777_777, a.n - 1)
    src = O.synth_entry(SEED, planted, max(int(counts[planted]), nq), 200)
    if a.no_match:
        src = O.synth_entry(SEED ^ 0x5555, 123, nq, 200)
    q = lb.Fingerprint.from_bools(src[:nq])
    for _ in range(3):
        c.query_key_device(q, key)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        c.query_key_device(q, key)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    r = {"n_entries": a.n, "records": total, "n_query": nq, "scan_ms": round(ms, 4),
         "algorithmic_GBps": round(25 * total / ms / 1e6, 1), "layout_GBps": round(32 * total / ms / 1e6, 1),
         "cells_per_s_G": round(nq * total / ms / 1e6, 2), "best": lb.Corpus.decode_key(int(key.item()) & (2**64 - 1))}
    print(json.dumps(r))
    out.append(r)
if a.json:
    json.dump(out, open(a.json, "w"), indent=1)
