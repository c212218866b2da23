#!/usr/bin/env python3
"""Randomized GPU-vs-oracle sweep of the ragged corpus (not part of the test suite): tools/fuzz_ragged.py [trials] [seed].
Corpora of 1..400 entries with 1..90 sub-fingerprints each (every length distribution: all short, all long, mixed,
a few very long ones past the record's saturating position fields), sub-fingerprint lengths 1..200, queries of
1..130 sub-fingerprints (shorter than, equal to and longer than the entries: the systolic kernel for short windows, the
task kernel's "A" and "B" passes and corpora that need both), every range, planted windows, duplicated entries (lowest index wins), empty
sub-fingerprints, 11 pairs; the per-entry scores (float bit patterns) and the top-1 against
oracle/lbad_oracle.c:lbo_corpus_best_ragged; every second trial also a batch of 2..11 queries of one or two lengths in one
call (round 5), every fifth through save / load and the sharded entry point."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lbaudiodetective_amd as lb
from oracle import oracle as O

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
batches = 0
t0 = time.time()
comm = lb.make_comm(0, 1)
tmp = tempfile.mkdtemp()
modes = {"short (systolic kernel)": 0, "A only": 0, "B only": 0, "A and B": 0}


def rand_fp(n, L, p_zero, p_both):
    pairs = (L + 1) // 2
    pos = rng.random((n, pairs)) < 0.5
    zero = rng.random((n, pairs)) < p_zero
    both = rng.random((n, pairs)) < p_both
    f = np.zeros((n, 2 * pairs), np.uint8)
    f[:, 0::2] = (pos & ~zero) | both
    f[:, 1::2] = (~pos & ~zero) | both
    return np.ascontiguousarray(f[:, :L])


for t in range(trials):
    L = int(rng.choice([1, 2, 3, 7, 31, 32, 33, 63, 64, 65, 127, 128, 129, 199, 200]))
    n = int(rng.integers(1, 400))
    shape = rng.integers(0, 5)
    if shape == 0:
        lens = rng.integers(1, 12, n)
    elif shape == 1:
        lens = rng.integers(40, 91, n)
    elif shape == 2:
        lens = rng.integers(1, 91, n)
    elif shape == 3:
        lens = np.full(n, int(rng.integers(1, 40)))
    else:
        lens = rng.integers(1, 30, n)
        lens[rng.integers(0, n)] = int(rng.choice([300, 4095, 4097, 5000]))
    p_zero, p_both = float(rng.choice([0.0, 0.02, 0.3])), float(rng.choice([0.0, 0.0, 0.05]))
    entries = [rand_fp(int(k), L, p_zero, p_both) for k in lens]
    for _ in range(int(rng.integers(0, 4))):                               # duplicates: ties go to the lowest index
        i, j = int(rng.integers(0, n)), int(rng.integers(0, n))
        entries[j] = entries[i].copy()
    if rng.integers(0, 4) == 0:
        entries[int(rng.integers(0, n))][:] = 0
    counts = np.array([e.shape[0] for e in entries], np.uint32)
    nq = int(rng.choice([1, 2, 3, 5, 6, 7, 8, 9, 10, 11, 12, 13, 21, 33, 48, 64, 65, 100, 130]))
    src = entries[int(rng.integers(0, n))]
    q = rand_fp(nq, L, p_zero, p_both)
    k = min(nq, src.shape[0])
    o = int(rng.integers(0, src.shape[0] - k + 1))
    q[:k] = src[o:o + k]
    if rng.integers(0, 2):
        flip = rng.random(q.shape) < 0.05
        q ^= flip.astype(np.uint8)
    rg = int(rng.choice([0, 1, 2, L // 2 + 1, L, L + 7]))
    look = min(nq, int(counts.max())) - 1
    modes["short (systolic kernel)" if (nq <= 7 or counts.max() <= 15) else "A only" if counts.min() > nq else "B only" if counts.max() <= nq else "A and B"] += 1
    corpus = lb.Corpus.ragged(L, n + 1, int(counts.sum()) + 200)
    if t % 3 == 1:
        corpus.set_kernel_variant(3)                                        # short entries split off to the systolic scan wherever a split exists
    flat = np.concatenate(entries, axis=0)
    packed = np.stack([lb.pack_subfingerprint(r) for r in flat]).view(np.uint8).reshape(-1, 32)
    cut = int(rng.integers(0, n + 1))                                       # appended in two pieces
    at = int(counts[:cut].sum())
    dev = torch.from_numpy(packed).cuda()
    if cut:
        corpus.append_ragged_packed_device(dev[:at], counts[:cut])
    if cut < n:
        corpus.append_ragged_packed_device(dev[at:], counts[cut:])
    fq = lb.Fingerprint.from_bools(q)
    bi, bs, want = O.corpus_best_ragged(q, (flat, counts), rg if rg else L, nthreads=8, want_scores=True)
    got = corpus.scores_device(fq, rg).cpu().numpy()
    top = corpus.query(fq, rg)
    ok = np.array_equal(got.view(np.uint32), want.view(np.uint32)) and top[0] == bi and \
        np.float32(top[1]).view(np.uint32) == np.float32(bs).view(np.uint32)
    if ok and t % 2 == 0:
        # round 5: a batch of 2..11 queries, one or two lengths mixed, in one call (groups of four / eight share a pass) --
        # every (index, score bits) against the oracle
        nb = int(rng.integers(2, 12))
        nq2 = int(rng.choice([1, 4, 7, 8, 10, 12, 16, 21, 33, 48, 70]))
        qs = []
        for b in range(nb):
            m = nq if (b % 3 or rng.integers(0, 2)) else nq2
            qq = rand_fp(m, L, p_zero, p_both)
            s2 = entries[int(rng.integers(0, n))]
            k2 = min(m, s2.shape[0])
            o2 = int(rng.integers(0, s2.shape[0] - k2 + 1))
            qq[:k2] = s2[o2:o2 + k2]
            qq ^= (rng.random(qq.shape) < 0.03).astype(np.uint8)
            qs.append(qq)
        batches = batches + 1
        gotb = corpus.query_batch([lb.Fingerprint.from_bools(x) for x in qs], rg)
        for x, g in zip(qs, gotb):
            wi, ws = O.corpus_best_ragged(x, (flat, counts), rg if rg else L, nthreads=8)
            if g[0] != wi or np.float32(g[1]).view(np.uint32) != np.float32(ws).view(np.uint32):
                ok = False
                print("BATCH MISMATCH", t, L, n, x.shape[0], rg, g, (wi, ws), flush=True)
    if ok and t % 5 == 0:
        p = os.path.join(tmp, "c.lbad")
        corpus.save(p)
        again = lb.Corpus.load(p, L, 0, n + 5)
        base = int(rng.integers(0, 1000))
        g2 = again.query_sharded(fq, comm, index_base=base, range_=rg)
        ok = g2 == ((bi + base) if bi >= 0 else -1, top[1])
    if not ok:
        bad += 1
        wrong = np.nonzero(got.view(np.uint32) != want.view(np.uint32))[0]
        print("RAGGED MISMATCH", t, L, n, shape, nq, rg, look, top, (bi, bs), wrong[:5], lens[wrong[:5]], flush=True)
print(f"{trials} trials, {bad} mismatches, {time.time() - t0:.1f} s; corpora by kernel path {modes}; {batches} batches of 2..11 queries")
sys.exit(1 if bad else 0)
