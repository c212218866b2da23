"""GPU parity of the ragged corpus (entries of any length, query of any length) against the CPU oracle.

The reference's corpus-shaped caller compares ONE file against ten of DIFFERENT lengths
(LBAudioDetectiveTests.m:57-91) and LBAudioDetectiveFingerprintCompareToFingerprint swaps and slides for any
n1 != n2 (LBAudioDetectiveFingerprint.m:119-149).  Scores are compared as float32 bit patterns, indices exactly.
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CSEED = 0x4C424145
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bits(x):
    return int(np.float32(x).view(np.uint32))


def _rand_fp(rng, n, L, p_zero=0.03, p_both=0.0):
    """n sub-fingerprints of L Booleans: sign pairs 10 / 01, some 00, optionally the 11 no extraction produces."""
    pairs = (L + 1) // 2
    pos = rng.random((n, pairs)) < 0.5
    zero = rng.random((n, pairs)) < p_zero
    both = rng.random((n, pairs)) < p_both
    f = np.zeros((n, 2 * pairs), np.uint8)
    f[:, 0::2] = (pos & ~zero) | both
    f[:, 1::2] = (~pos & ~zero) | both
    return np.ascontiguousarray(f[:, :L])


def _ragged_corpus(lb, gpu, entries, L):
    counts = np.array([e.shape[0] for e in entries], np.uint32)
    c = lb.Corpus.ragged(L, len(entries), int(counts.sum()))
    flat = np.concatenate(entries, axis=0)
    packed = np.zeros((flat.shape[0], 8), np.uint32)
    for i, row in enumerate(flat):
        packed[i] = lb.pack_subfingerprint(row)
    c.append_ragged_packed_device(gpu.from_numpy(packed.view(np.uint8).reshape(-1, 32)).cuda(), counts)
    return c, counts


def _check_query(lb, oracle, corpus, entries, q, rg):
    bi, bs, want = oracle.corpus_best_ragged(q, entries, rg if rg else q.shape[1], want_scores=True)
    fq = lb.Fingerprint.from_bools(q)
    got = corpus.scores_device(fq, rg).cpu().numpy()
    bad = np.nonzero(got.view(np.uint32) != want.view(np.uint32))[0]
    assert bad.size == 0, (q.shape, rg, bad[:5], got[bad[:5]], want[bad[:5]], [entries[i].shape[0] for i in bad[:5]])
    idx, score = corpus.query(fq, rg)
    assert idx == bi and np.float32(score).view(np.uint32) == np.float32(bs).view(np.uint32), (idx, score, bi, bs)


@pytest.mark.parametrize("L", [200, 199, 33, 2, 1, 64])
def test_ragged_random_shapes(lb, gpu, oracle, L):
    """Random entries of 1..70 sub-fingerprints against queries shorter, equal and longer, all ranges; 11 pairs
    and empty sub-fingerprints included; every entry's score and the top-1 equal the oracle's."""
    rng = np.random.default_rng(1000 + L)
    n_entries = 300
    lens = rng.integers(1, 71, n_entries)
    lens[:8] = [1, 2, 70, 21, 21, 20, 22, 64]
    entries = [_rand_fp(rng, int(n), L, p_zero=0.05, p_both=0.02) for n in lens]
    entries[9][:] = 0                                                    # an entry without any set Boolean
    entries[10][1::2] = 0
    corpus, _ = _ragged_corpus(lb, gpu, entries, L)
    assert len(corpus) == n_entries and corpus.subfingerprint_total == int(lens.sum())
    # (7 / 8: where the systolic scan of short queries hands over to the task kernel)
    for nq in (1, 2, 5, 7, 8, 12, 15, 16, 21, 48, 64, 70):
        q = _rand_fp(rng, nq, L, p_zero=0.05, p_both=0.02)
        src = entries[int(rng.integers(0, n_entries))]
        k = min(nq, src.shape[0])
        q[:k] = src[:k]                                                  # partly a copy of some entry
        q[::2, : max(1, L // 5)] ^= 1
        for rg in sorted({0, L, max(1, L // 2), 1, 7, L + 50, max(1, L - 1)}):
            _check_query(lb, oracle, corpus, entries, q, rg)


@pytest.mark.parametrize("longest", [7, 15, 16])
def test_ragged_short_entries_against_any_query(lb, gpu, oracle, longest):
    """A corpus whose longest entry has at most 15 sub-fingerprints goes through the systolic scan whatever the query's
    length (one record per lane up to 7, four above); with a longest entry of 16 the task kernel takes over."""
    rng = np.random.default_rng(4000 + longest)
    lens = rng.integers(1, longest + 1, 500)
    lens[:3] = [longest, 1, max(1, longest - 1)]
    entries = [_rand_fp(rng, int(n), 200, p_zero=0.03, p_both=0.01) for n in lens]
    corpus, _ = _ragged_corpus(lb, gpu, entries, 200)
    for nq in (1, 6, 7, 8, 12, 15, 16, 17, 40, 100):
        q = _rand_fp(rng, nq, 200, p_zero=0.03, p_both=0.01)
        src = entries[int(rng.integers(0, len(entries)))]
        at = int(rng.integers(0, max(1, nq - src.shape[0] + 1)))
        k = min(nq - at, src.shape[0])
        q[at:at + k] = src[:k]
        q[::3, :20] ^= 1
        for rg in (0, 64, 1):
            _check_query(lb, oracle, corpus, entries, q, rg)
    qs = [_rand_fp(rng, 12, 200) for _ in range(9)]
    got = corpus.query_batch([lb.Fingerprint.from_bools(q) for q in qs])
    for q, g in zip(qs, got):
        bi, bs = oracle.corpus_best_ragged(q, entries, 200)
        assert (g[0], _bits(g[1])) == (bi, _bits(bs)), (longest, g, bi, bs)


@pytest.mark.parametrize("variant", [0, 3, 4])
def test_ragged_scan_split_between_the_two_kernels(lb, gpu, oracle, variant):
    """A corpus that mixes many entries of at most 15 sub-fingerprints with longer ones, asked queries longer than the short
    entries: the library may hand the short entries to the systolic scan and keep the rest on the task kernel (two launches
    that max into the same keys and scores; variant 3 forces that, 4 forbids it, 0 decides by the entries' lengths) --
    every entry's score and the top-1 equal the oracle's either way, also for batches and planted matches on both sides."""
    rng = np.random.default_rng(5100 + variant)
    lens = np.concatenate([rng.integers(1, 16, 700), rng.integers(16, 90, 150)])
    rng.shuffle(lens)
    lens[:4] = [15, 16, 1, 89]
    entries = [_rand_fp(rng, int(n), 200, p_zero=0.03, p_both=0.01) for n in lens]
    corpus, _ = _ragged_corpus(lb, gpu, entries, 200)
    corpus.set_kernel_variant(variant)
    short = [i for i, n in enumerate(lens) if n <= 15]
    long_ = [i for i, n in enumerate(lens) if n > 40]
    for nq in (16, 17, 30, 48, 100):
        for src_i in (short[int(rng.integers(0, len(short)))], long_[int(rng.integers(0, len(long_)))]):
            q = _rand_fp(rng, nq, 200, p_zero=0.03, p_both=0.01)
            src = entries[src_i]
            k = min(nq, src.shape[0])
            at = int(rng.integers(0, nq - k + 1))
            q[at:at + k] = src[:k]                                       # the entry (or its beginning) planted inside the query
            q[::5, :10] ^= 1
            for rg in (0, 64):
                _check_query(lb, oracle, corpus, entries, q, rg)
    for nq in (21, 60):
        qs = [_rand_fp(rng, nq, 200) for _ in range(7)]
        for i, q in enumerate(qs):
            src = entries[short[i]] if i % 2 else entries[long_[i]]
            k = min(nq, src.shape[0])
            q[:k] = src[:k]
        got = corpus.query_batch([lb.Fingerprint.from_bools(q) for q in qs])
        for q, g in zip(qs, got):
            bi, bs = oracle.corpus_best_ragged(q, entries, 200)
            assert (g[0], _bits(g[1])) == (bi, _bits(bs)), (variant, nq, g, bi, bs)


def test_ragged_long_query_against_long_entries(lb, gpu, oracle):
    """Both sides long (a window reaches back 64 records or more): the per-entry kernel; and entries beyond the
    saturation of the record's 12-bit position fields."""
    rng = np.random.default_rng(77)
    lens = [5000, 90, 64, 65, 4096, 200, 1, 130]
    entries = [_rand_fp(rng, n, 200) for n in lens]
    corpus, _ = _ragged_corpus(lb, gpu, entries, 200)
    # (480 / 481: the longest query whose block lives in LDS next to 131 KB of tables, queues and staging blocks, and the
    # first that is read through the scalar cache; 108 / 109 and 217 / 218: where batches go from four to two to one query per pass)
    for nq in (64, 65, 100, 130, 300, 5, 48, 480, 481):
        q = _rand_fp(rng, nq, 200)
        src = entries[0]
        k = min(nq, 60)
        q[:k] = src[4500:4500 + k]
        for rg in (0, 64):
            _check_query(lb, oracle, corpus, entries, q, rg)
    for nq in (108, 109, 217, 218):
        qs = [_rand_fp(rng, nq, 200) for _ in range(5)]
        for i, q in enumerate(qs):
            q[:50] = entries[0][700 * i + 3:700 * i + 53]
        got = corpus.query_batch([lb.Fingerprint.from_bools(q) for q in qs])
        for q, g in zip(qs, got):
            bi, bs = oracle.corpus_best_ragged(q, entries, 200)
            assert (g[0], _bits(g[1])) == (bi, _bits(bs)), (nq, g, bi, bs)


def test_ragged_ties_zero_and_append(lb, gpu, oracle):
    """Lowest index wins ties across entries of different lengths (T.m:80 strict '<'); a corpus that scores 0
    selects nothing; entries appended one fingerprint at a time and in batches give the same corpus."""
    rng = np.random.default_rng(5)
    a = _rand_fp(rng, 30, 200)
    entries = [_rand_fp(rng, 12, 200), a.copy(), _rand_fp(rng, 40, 200), np.concatenate([_rand_fp(rng, 7, 200), a]), a.copy()]
    c = lb.Corpus.ragged(200, 16, 400)
    for e in entries[:2]:
        c.append_fingerprint(lb.Fingerprint.from_bools(e))
    rest = entries[2:]
    packed = np.stack([lb.pack_subfingerprint(r) for e in rest for r in e]).view(np.uint8).reshape(-1, 32)
    c.append_ragged_packed_device(gpu.from_numpy(packed).cuda(), [e.shape[0] for e in rest])
    q = a[5:26]
    assert c.query(lb.Fingerprint.from_bools(q)) == (1, 1.0)
    _check_query(lb, oracle, c, entries, q, 0)
    zero = np.zeros((3, 200), np.uint8)
    assert c.query(lb.Fingerprint.from_bools(zero)) == (-1, 0.0)
    with pytest.raises(lb.LBAudioDetectiveError):
        c.append_packed_device(gpu.zeros((1, 5, 32), dtype=gpu.uint8, device="cuda"))   # uniform append on a ragged corpus
    with pytest.raises(lb.LBAudioDetectiveError):
        lb.Corpus.ragged(256, 4, 16)                                                    # length above 200


def test_ragged_save_load(lb, gpu, oracle, tmp_path):
    rng = np.random.default_rng(9)
    entries = [_rand_fp(rng, int(n), 200) for n in rng.integers(1, 60, 50)]
    c, _ = _ragged_corpus(lb, gpu, entries, 200)
    p = str(tmp_path / "ragged.lbad")
    c.save(p)
    d = lb.Corpus.load(p, 200, 0, 80)
    assert len(d) == 50 and d.subfingerprint_total == c.subfingerprint_total
    q = _rand_fp(rng, 21, 200)
    q[:15] = entries[33][2:17]
    _check_query(lb, oracle, d, entries, q, 0)
    d.append_fingerprint(lb.Fingerprint.from_bools(q))                   # a loaded corpus keeps growing
    assert d.query(lb.Fingerprint.from_bools(q)) == (50, 1.0)


def _record_words(row, L, old_layout):
    """One sub-fingerprint of L Booleans as the eight words of a corpus-file record (k_sliding.hip): round-4 layout
    or the round-3 one ("LBADCRP2": P at bits 0..99, N at bits 100..199, place fields above)."""
    pairs = (L + 1) // 2
    P = sum(int(row[2 * p]) << p for p in range(pairs))
    N = sum(int(row[2 * p + 1]) << p for p in range(pairs) if 2 * p + 1 < L)
    if old_layout:
        v = P | (N << 100) | (0xABC << 200) | (0xDEF << 212) | (0x12345678 << 224)     # place fields: garbage on purpose
        return [(v >> (32 * w)) & 0xFFFFFFFF for w in range(8)]
    w = [(P >> (32 * k)) & 0xFFFFFFFF for k in range(4)] + [(N >> (32 * k)) & 0xFFFFFFFF for k in range(4)]
    return w


def test_bound_pruning_of_top1_scans_is_exact(lb, gpu, oracle):
    """Top-1 scans of a ragged corpus drop groups of offsets that cannot reach the best match published so far (round 4):
    with strong matches planted at several places (early, late, twice with equal scores -- the lower index must win -- and
    none at all) the key equals the unpruned scan's and the oracle's best entry, for queries shorter and longer than the
    entries around them."""
    rng = np.random.default_rng(17)
    SEED = 0x4C424145
    n = 60_000
    counts = oracle.synth_ragged_counts(SEED, 0, n, 20, 70)
    flat = oracle.synth_ragged_entries(SEED, 0, counts, 200)
    starts = np.concatenate([[0], np.cumsum(counts)])
    for nq, plants in ((21, [50_000]), (21, [5]), (21, [31_000, 44_000]), (33, [59_999, 10]), (60, [40_000, 123]), (75, [7]), (21, [])):
        src = oracle.synth_entry(SEED ^ 77, 1, nq, 200)
        ent = flat.copy()
        for e in plants:
            ne = int(counts[e])
            k = min(nq, ne)
            at = int(starts[e]) + int(rng.integers(0, ne - k + 1))
            ent[at:at + k] = src[:k]                            # the same window in every planted entry: equal scores where the lengths allow
        corpus = lb.Corpus.ragged(200, n, int(counts.sum()))
        packed = np.stack([lb.pack_subfingerprint(r) for r in ent]).view(np.uint8).reshape(-1, 32)
        corpus.append_ragged_packed_device(gpu.from_numpy(packed).cuda(), counts)
        q = lb.Fingerprint.from_bools(src)
        corpus.set_bound_pruning(True)
        pruned = corpus.query(q)
        corpus.set_bound_pruning(False)
        full = corpus.query(q)
        want_i, want_s, _ = oracle.corpus_best_ragged(src, (ent, counts), 200, nthreads=8, want_scores=True)
        assert pruned == full, (nq, plants, pruned, full)
        assert pruned[0] == want_i and np.float32(pruned[1]).view(np.uint32) == np.float32(want_s).view(np.uint32), (nq, plants, pruned, want_i, want_s)
        if plants:
            assert pruned[1] > 0.7, (plants, pruned)
        corpus.dispose()


def test_corpus_file_records_are_not_trusted(lb, gpu, oracle, tmp_path):
    """Round-3 advice: a corpus file's records carried index fields the scan wrote through.  Now nothing a record
    carries besides its Booleans survives the loader: a file with garbage in every reserved / derived bit and in the
    pairs beyond the length, and a round-3 file with nonsense place fields, load to corpora whose every score equals
    the oracle's; a file whose counts do not add up is refused."""
    import struct
    rng = np.random.default_rng(31)
    for L in (200, 33):
        lens = rng.integers(1, 40, 30)
        entries = [_rand_fp(rng, int(n), L, p_zero=0.05, p_both=0.02) for n in lens]
        flat = np.concatenate(entries, axis=0)
        for old_layout, magic in ((False, b"LBADCRP3"), (True, b"LBADCRP2")):
            recs = np.array([_record_words(r, L, old_layout) for r in flat], np.uint32)
            if not old_layout:
                pairs = (L + 1) // 2
                junk = rng.integers(0, 2**32, recs.shape, dtype=np.uint32)
                keep = np.zeros(8, np.uint32)                            # bits that hold pairs < pairs
                for k in range(4):
                    nbits = min(32, max(0, pairs - 32 * k)) if k < 3 else min(4, max(0, pairs - 96))
                    keep[k] = keep[4 + k] = (1 << nbits) - 1 if nbits < 32 else 0xFFFFFFFF
                recs = (recs & keep) | (junk & ~keep)                     # row field, reserved bits, pairs beyond L: junk
            p = str(tmp_path / f"crafted_{L}_{int(old_layout)}.lbad")
            with open(p, "wb") as f:
                f.write(magic + struct.pack("<IIQQ", L, 0, len(entries), flat.shape[0]))
                f.write(np.asarray(lens, np.uint32).tobytes())
                f.write(recs.tobytes())
            c = lb.Corpus.load(p, L, 0, 0)
            assert len(c) == len(entries) and c.subfingerprint_total == flat.shape[0]
            for nq in (1, 7, 21, 45):
                q = _rand_fp(rng, nq, L, p_zero=0.05, p_both=0.02)
                k = min(nq, entries[3].shape[0])
                q[:k] = entries[3][:k]
                for rg in (0, max(1, L // 2)):
                    _check_query(lb, oracle, c, entries, q, rg)
        bad = str(tmp_path / "bad_counts.lbad")
        with open(bad, "wb") as f:
            wrong = np.asarray(lens, np.uint32).copy()
            wrong[0] += 1
            f.write(b"LBADCRP3" + struct.pack("<IIQQ", L, 0, len(entries), flat.shape[0]) + wrong.tobytes() + recs.tobytes())
        with pytest.raises(Exception):
            lb.Corpus.load(bad, L, 0, 0)


def test_birds_ten_archives_in_one_corpus(lb, gpu, oracle):
    """Upstream's best-match loop as corpus queries on its own fixtures (LBAudioDetectiveTests.m:57-91).  The ten
    archive recordings (48 .. 115 sub-fingerprints each) live in ONE corpus and each of the 50 sequences is one
    query whose per-entry scores are a column of the test's 10 x 10 matrix; then the literal arrangement: the ten
    sequences of a test are the corpus and each original is the query, top-1 = the test's `maxMatch` / `failed`.
    Both equal the pairwise LBAudioDetectiveFingerprintCompareToFingerprint calls and the oracle bit for bit."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import birds_matrix as bm
    suffixes = [bm.ESSAY["tests"][t]["suffix"] for t in bm.TESTS]
    names = bm.BIRDS + [b + s for s in suffixes for b in bm.BIRDS]
    fps = bm.fingerprints_gpu(names, 1, 1, 0)
    archives = [fps[b] for b in bm.BIRDS]
    assert len({a.shape[0] for a in archives}) > 1                       # really of different lengths
    corpus, _ = _ragged_corpus(lb, gpu, archives, 200)
    pairwise = bm.matrices("gpu")                                        # 100 pairwise calls per test
    for t, s in zip(bm.TESTS, suffixes):
        seqs = [fps[b + s] for b in bm.BIRDS]
        m = np.zeros((10, 10), np.float32)
        for j, q in enumerate(seqs):                                     # 50 queries in all
            fq = lb.Fingerprint.from_bools(q)
            m[:, j] = corpus.scores_device(fq).cpu().numpy()
            bi, bs, want = oracle.corpus_best_ragged(q, archives, 200, want_scores=True)
            assert np.array_equal(m[:, j].view(np.uint32), want.view(np.uint32)), (t, j)
            assert corpus.query(fq) == (bi, bs)
        assert np.array_equal(m.astype(np.float64) * 100.0, pairwise[t]), t
        seq_corpus, _ = _ragged_corpus(lb, gpu, seqs, 200)
        for i, a in enumerate(archives):
            want_i = int(np.argmax(pairwise[t][i])) if pairwise[t][i].max() > 0 else -1
            idx, score = seq_corpus.query(lb.Fingerprint.from_bools(a))
            assert idx == want_i and np.float64(np.float32(score)) * 100.0 == pairwise[t][i].max(), (t, i)
    assert bm.check(pairwise) == []


@pytest.mark.parametrize("n", [1_000_000])
def test_full_size_ragged_corpus(lb, gpu, oracle, n):
    """1 M synthetic entries of 20..70 sub-fingerprints (45 M records, 1.44 GB), query of 21 cut out of entry
    777 777 with 7 % of its pairs flipped.  Bit-equal to the oracle on a 20 000-entry sample of per-entry scores;
    at full size: the planted entry wins, a twin appended behind it loses the tie to the lower index, shards
    queried separately reduce to the same key."""
    counts = oracle.synth_ragged_counts(CSEED, 0, n, 20, 70)
    assert all(int(counts[i]) == oracle.lib().lbo_synth_ragged_count(CSEED, i, 20, 70) for i in (0, 1, 777_777, n - 1))
    total = int(counts.sum())
    packed = lb.synth_ragged_corpus_device(CSEED, 0, counts, 200)
    corpus = lb.Corpus.ragged(200, n + 1, total + 80)
    corpus.append_ragged_packed_device(packed, counts)
    planted = 777_777
    src = oracle.synth_entry(CSEED, planted, int(counts[planted]), 200)
    q = src[3:24].copy()
    rng = np.random.default_rng(3)
    flip = rng.random((21, 100)) < 0.07
    pos = q[:, 0::2].copy()
    q[:, 0::2] = np.where(flip, q[:, 1::2], pos)
    q[:, 1::2] = np.where(flip, pos, q[:, 1::2])
    fq = lb.Fingerprint.from_bools(q)
    scores = corpus.scores_device(fq).cpu().numpy()
    # sample: the first 10 000, the planted entry's neighbourhood, the last 5 000
    for lo, hi in ((0, 10_000), (planted - 2_500, planted + 2_500), (n - 5_000, n)):
        ent = oracle.synth_ragged_entries(CSEED, lo, counts[lo:hi], 200)
        _, _, want = oracle.corpus_best_ragged(q, (ent, counts[lo:hi]), 200, nthreads=8, want_scores=True)
        assert np.array_equal(scores[lo:hi].view(np.uint32), want.view(np.uint32)), (lo, hi)
    idx, score = corpus.query(fq)
    assert idx == planted and np.float32(score).view(np.uint32) == scores[planted].view(np.uint32)
    assert int(np.argmax(scores)) == planted and 0.9 < score < 0.96
    # twin of the planted entry behind everything: equal score, the lower index wins (T.m:80)
    corpus.append_fingerprint(lb.Fingerprint.from_bools(src))
    assert corpus.query(fq) == (planted, score)
    # four shards of the same data, keys max-reduced like the sharded query does
    off = np.concatenate([[0], np.cumsum(counts, dtype=np.int64)])
    keys = gpu.zeros(4, dtype=gpu.int64, device="cuda")
    bounds = [0, n // 4, n // 2, 3 * n // 4, n]
    shards = []
    for r in range(4):
        lo, hi = bounds[r], bounds[r + 1]
        sh = lb.Corpus.ragged(200, hi - lo, int(off[hi] - off[lo]))
        sh.append_ragged_packed_device(packed[int(off[lo]):int(off[hi])], counts[lo:hi])
        sh.query_key_device(fq, keys[r:r + 1], index_base=lo)
        shards.append(sh)
    gpu.cuda.synchronize()
    best = max(int(k) & 0xFFFFFFFFFFFFFFFF for k in keys.cpu().numpy().astype(np.uint64))
    assert lb.Corpus.decode_key(best) == (planted, score)


# ---------------------------------------------------------------------------------------------
# the exchange step inside the library: RCCL behind the C ABI
# ---------------------------------------------------------------------------------------------
def test_sharded_query_through_native_rccl(lb, gpu, oracle):
    """LBAudioDetectiveCorpusQuerySharded at world size 1 through a REAL ncclCommInitRank (the library's own
    dlopen-ed RCCL): scan + ncclAllReduce(ncclUint64, ncclMax) + read-back equals the plain query, for the uniform
    and the ragged corpus, single and batched; the 32-bit global index is guarded."""
    comm = lb.make_comm(0, 1)
    n = 50_000
    host = oracle.synth_corpus(CSEED, 0, 2000, 5, 200)
    c = lb.Corpus(200, 5, n)
    c.append_packed_device(lb.synth_corpus_device(CSEED, 0, n, 5, 200))
    qs = []
    for probe in (7, 1234, 1999):
        q = host[probe].copy()
        q[:, :14] ^= 1
        qs.append(lb.Fingerprint.from_bools(q))
    for fq in qs:
        want = c.query(fq)
        assert c.query_sharded(fq, comm) == want
        base = 1_000_000
        gi, gs = c.query_sharded(fq, comm, index_base=base)
        assert (gi - base, gs) == want
    assert c.query_batch_sharded(qs, comm) == [c.query(f) for f in qs]
    with pytest.raises(lb.LBAudioDetectiveError):
        c.query_sharded(qs[0], comm, index_base=2**32 - n + 1)          # index_base + count > 2^32
    assert c.query_sharded(qs[0], comm, index_base=2**32 - n)[0] == 2**32 - n + 7
    # ragged corpus, query of another length
    rng = np.random.default_rng(21)
    entries = [_rand_fp(rng, int(k), 200) for k in rng.integers(1, 60, 400)]
    r, _ = _ragged_corpus(lb, gpu, entries, 200)
    q = _rand_fp(rng, 21, 200)
    q[:18] = entries[123][1:19] if entries[123].shape[0] >= 19 else q[:18]
    fq = lb.Fingerprint.from_bools(q)
    assert r.query_sharded(fq, comm, index_base=10) == (r.query(fq)[0] + 10, r.query(fq)[1])
    # the Python sharded wrapper on the native path
    sc = lb.ShardedCorpus(200, 5, n, rank=0, world_size=1, comm=comm)
    sc.append_packed_device(lb.synth_corpus_device(CSEED, 0, n, 5, 200))
    assert sc.query(qs[1]) == c.query(qs[1])
    comm.dispose()


def test_sharded_c_host_one_rank(lb, gpu, tmp_path):
    """examples/sharded_query.c as its own process: a pure-C rank (no Python, no torch) builds its shard, creates
    the communicator through the library and finds the planted entry."""
    import subprocess
    sys.path.insert(0, os.path.dirname(__file__))
    from test_capi import _build_example
    exe = _build_example(tmp_path, lb, "sharded_query")
    out = subprocess.run([exe, "0", "1", str(tmp_path / "rccl.id"), "3000000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "best match index 1777777 " in out.stdout and "score 0.919" in out.stdout, out.stdout


def test_sharded_entry_point_with_two_ranks_in_one_process(lb, gpu, oracle):
    """LBAudioDetectiveCorpusQueryBatchShardedWith -- the C entry point of the sharded query with the exchange step
    handed in -- driven by TWO "ranks" inside this process: two threads, each with its own shard of the corpus on the
    one GPU and its own stream, and a collective written here (copy the keys to the host, meet at a barrier, MAX, copy
    back) where a multi-GPU run has ncclAllReduce(ncclUint64, ncclMax).  RCCL refuses two ranks on one device, the
    reduction rule does not care where the ranks live.  Asserted: every rank receives the global best match; an entry
    present in BOTH shards resolves to the LOWER global index (LBAudioDetectiveTests.m:80-83 across ranks); a rank
    whose own scan cannot run (no corpus; indices beyond 2^32) still joins the exchange, the other rank gets its own
    result and does not hang; a batch larger than one exchange's key block is cut the same way on both ranks."""
    import ctypes as C
    import threading
    from lbaudiodetective_amd import _native as N
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    rng = np.random.default_rng(41)
    n0, n1 = 300, 260
    ent = [_rand_fp(rng, int(n), 200) for n in rng.integers(3, 40, n0 + n1)]
    twin = _rand_fp(rng, 25, 200)
    ent[123] = twin.copy()                       # global index 123 (rank 0)
    ent[n0 + 17] = twin.copy()                   # global index 317 (rank 1): same score, must lose the tie
    shards = [_ragged_corpus(lb, gpu, ent[:n0], 200)[0], _ragged_corpus(lb, gpu, ent[n0:], 200)[0]]
    bases = [0, n0]
    queries = [twin[2:23], ent[n0 + 100][:3] if ent[n0 + 100].shape[0] >= 3 else ent[n0 + 100], _rand_fp(rng, 9, 200), ent[5][:1]]
    want = [oracle.corpus_best_ragged(q, ent, 200) for q in queries]
    assert want[0] == (123, 1.0) and want[1][0] >= n0

    barrier = threading.Barrier(2)
    slots = [None, None]
    calls = [0, 0]

    def make_collective(rank):
        def all_reduce(context, keys, count, stream):
            hip.hipStreamSynchronize(stream)
            host = (C.c_uint64 * count)()
            hip.hipMemcpy(host, keys, 8 * count, 2)                      # device to host
            slots[rank] = list(host)
            barrier.wait(timeout=60)
            best = [max(a, b) for a, b in zip(slots[0], slots[1])]
            barrier.wait(timeout=60)
            out = (C.c_uint64 * count)(*best)
            hip.hipMemcpy(keys, out, 8 * count, 1)                      # host to device
            calls[rank] += 1
            return 0
        return N.AllReduceMaxFn(all_reduce)

    def run(rank, corpus_of_rank, base_of_rank, fps, out):
        gpu.cuda.set_device(0)
        stream = gpu.cuda.Stream()
        try:
            out[rank] = ("ok", corpus_of_rank.query_batch_sharded_with(fps, make_collective(rank), index_base=base_of_rank, stream=stream))
        except lb.LBAudioDetectiveError as e:
            out[rank] = ("error", e.status)

    def both(corpora, base_list, fps):
        out = [None, None]
        th = [threading.Thread(target=run, args=(r, corpora[r], base_list[r], fps, out)) for r in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=120)
        assert not any(t.is_alive() for t in th), "a rank hangs in the exchange"
        return out

    fps = [lb.Fingerprint.from_bools(q) for q in queries]
    out = both(shards, bases, fps)
    assert out[0] == out[1] == ("ok", want), (out, want)
    # a rank whose indices do not fit the key (index base + count > 2^32) joins with empty keys and reports its error;
    # the other rank's result is the best match of ITS shard alone
    out = both(shards, [2**32 - 10, n0], fps[:2])
    alone = [oracle.corpus_best_ragged(q, ent[n0:], 200) for q in queries[:2]]
    assert out[0] == ("error", 1) and out[1] == ("ok", [(i + n0 if i >= 0 else -1, s) for i, s in alone]), out
    # a rank without a corpus at all (the C entry point takes NULL): same thing
    class NoCorpus:
        _L = shards[0]._L
        _ref = None
        query_batch_sharded_with = lb.Corpus.query_batch_sharded_with
    out = both([shards[0], NoCorpus()], bases, fps[:1])
    assert out[1] == ("error", 1) and out[0] == ("ok", [oracle.corpus_best_ragged(queries[0], ent[:n0], 200)]), out
    # more queries than one exchange carries: two exchanges on both ranks, the same results
    before = list(calls)
    many = [fps[i % len(fps)] for i in range(lb.SHARD_KEYS + 5)]
    out = both(shards, bases, many)
    assert out[0] == out[1] and out[0][0] == "ok" and out[0][1] == [want[i % len(fps)] for i in range(len(many))]
    assert [c - b for c, b in zip(calls, before)] == [2, 2]


def test_bench_collective_paths_on_nccl_at_one_rank(gpu):
    """Every branch bench.py takes at N > 1 that is well defined at one rank, on the REAL backend: `torch.distributed.run
    --nproc-per-node=1 bench.py --gpus 1 --force-dist` initialises the nccl (= RCCL) process group on this GPU, gathers
    the device descriptions, MAX-reduces passes / elapsed / stats, broadcasts the library communicator's id and the
    query, gathers packed fingerprints, runs the sharded query through ncclAllReduce inside the library, and tears the
    group down.  The bench is a CHILD process (a process that has touched the GPU must never exec another program)."""
    import json
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--clips", "2000",
           "--corpus", "200000", "--corpus-hbm", "0", "--steps", "2", "--warmup", "1", "--min-seconds", "0.2",
           "--no-cpu-baseline", "--no-other-configs", "--no-sliding", "--no-files"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 1 and r["rccl_ranks"] == 1 and r["backend"] == "nccl" and "force_dist" in r
    assert r["devices"][0]["rank"] == 0 and r["value"] > 0
    assert r["gather_packed"] == {"clips_per_rank": 16, "gathered": 16}
    c = r["compare"]
    assert c["collective_fallback"] is False and c["found_planted"] is True and c["allreduce_ms"] is not None
    assert r["parity"]["bit_exact"] is True if "parity" in r else True
    assert r["self_check"]["ok"]


@pytest.mark.parametrize("L,seed", [(200, 1), (199, 2), (64, 3)])
def test_ragged_batches_equal_single_queries(lb, gpu, oracle, L, seed):
    """Round 5: several queries of ONE length share their passes over the records (four per launch of the task scan, eight of
    the scan of short queries -- since round 6 compare_short_multi_kernel, lengths 1..12); a batch of mixed lengths runs a group per length.  Upstream's caller is Q x N
    (LBAudioDetectiveTests.m:57-91: ten originals against ten candidates).  Every (index, score bits) of a batch equals the
    single query's and the oracle's, whatever the order and the group sizes (1, 2, 4, 8 and what is left over)."""
    rng = np.random.default_rng(4200 + seed)
    n_entries = 400
    lens = rng.integers(1, 71, n_entries)
    lens[:6] = [70, 21, 22, 20, 48, 1]
    entries = [_rand_fp(rng, int(n), L, p_zero=0.05, p_both=0.02) for n in lens]
    corpus, _ = _ragged_corpus(lb, gpu, entries, L)

    def make(nq):
        q = _rand_fp(rng, nq, L, p_zero=0.05, p_both=0.02)
        src = entries[int(rng.integers(0, n_entries))]
        k = min(nq, src.shape[0])
        q[:k] = src[:k]
        q[::3, : max(1, L // 7)] ^= 1
        return q

    # (round 6: batches of queries of up to 12 sub-fingerprints go through compare_short_multi_kernel -- every length 1..12,
    # groups of 8 / 4 / 2 and a single left over; 13 is the first length of the task kernel's batches again)
    for lengths in ([21] * 4, [21] * 8, [48] * 7, [5] * 8, [5] * 3 + [12] * 9, [21, 48, 21, 5, 48, 21, 21, 64, 21, 2, 70, 21, 48],
                    [30, 30], [16] * 5, [33] * 11, [1] * 2 + [2] * 3 + [3] * 4 + [4] * 2, [6] * 8 + [7] * 7, [8] * 15, [9] * 4 + [10] * 6,
                    [11] * 2 + [12] * 8 + [13] * 8):
        qs = [make(nq) for nq in lengths]
        fps = [lb.Fingerprint.from_bools(q) for q in qs]
        for rg in (0, max(1, L // 2), 7):
            got = corpus.query_batch(fps, rg)
            for q, fq, g in zip(qs, fps, got):
                bi, bs = oracle.corpus_best_ragged(q, entries, rg if rg else L)
                one = corpus.query(fq, rg)
                assert (g[0], _bits(g[1])) == (one[0], _bits(one[1])) == (bi, _bits(bs)), (lengths, rg, q.shape, g, one, bi, bs)
    # the same through the device form, with an index base, on a stream of its own
    s = gpu.cuda.Stream()
    keys = gpu.zeros(8, dtype=gpu.int64, device="cuda")
    qs = [make(21) for _ in range(8)]
    fps = [lb.Fingerprint.from_bools(q) for q in qs]
    with gpu.cuda.stream(s):
        corpus.query_batch_keys_device(fps, keys, 0, index_base=1000, stream=s)
    s.synchronize()
    for q, k in zip(qs, keys.tolist()):
        bi, bs = oracle.corpus_best_ragged(q, entries, L)
        idx, sc = lb.Corpus.decode_key(k)
        assert (idx, _bits(sc)) == (bi + 1000, _bits(bs))


def test_full_size_short_query_batches(lb, gpu, oracle):
    """Round 6, compare_short_multi_kernel at BASELINE size: 1 M synthetic entries of 4..70 sub-fingerprints (37 M records), eight
    queries of 5 and eight of 11 in one call each -- cut out of entries spread over the corpus (one of them an entry of exactly
    the query's length, which the batch kernel leaves to the systolic scan's second launch), a few sign pairs flipped.  Every
    key of a batch equals the single query's (another kernel) and the planted entry; around every planted entry the per-entry
    scores of the single query equal the oracle's bit for bit; a twin appended behind everything loses the tie."""
    n = 1_000_000
    counts = oracle.synth_ragged_counts(CSEED, 0, n, 4, 70)
    total = int(counts.sum())
    packed = lb.synth_ragged_corpus_device(CSEED, 0, counts, 200)
    corpus = lb.Corpus.ragged(200, n + 8, total + 200)
    corpus.append_ragged_packed_device(packed, counts)
    del packed
    rng = np.random.default_rng(66)
    for nq in (5, 11):
        exact = int(np.nonzero(counts == nq)[0][3])                      # an entry of exactly nq sub-fingerprints ("B": not longer)
        homes = [123, 100_777, 250_001, exact, 499_999, 640_000, 777_777, n - 2]
        qs, planted = [], []
        for e in homes:
            while counts[e] < nq:
                e += 1
            src = oracle.synth_entry(CSEED, e, int(counts[e]), 200)
            o = int(rng.integers(0, counts[e] - nq + 1))
            q = src[o:o + nq].copy()
            flip = rng.random((nq, 100)) < 0.04
            pos = q[:, 0::2].copy()
            q[:, 0::2] = np.where(flip, q[:, 1::2], pos)
            q[:, 1::2] = np.where(flip, pos, q[:, 1::2])
            qs.append(q)
            planted.append(e)
        fps = [lb.Fingerprint.from_bools(q) for q in qs]
        got = corpus.query_batch(fps)
        for q, fq, e, g in zip(qs, fps, planted, got):
            one = corpus.query(fq)
            assert (g[0], _bits(g[1])) == (one[0], _bits(one[1])), (nq, e, g, one)
            assert g[0] == e and g[1] > 0.9, (nq, e, g)
            lo, hi = max(0, e - 300), min(n, e + 300)
            ent = oracle.synth_ragged_entries(CSEED, lo, counts[lo:hi], 200)
            bi, bs, want = oracle.corpus_best_ragged(q, (ent, counts[lo:hi]), 200, nthreads=8, want_scores=True)
            assert lo + bi == e and _bits(bs) == _bits(g[1])
            scores = corpus.scores_device(fq).cpu().numpy()
            assert np.array_equal(scores[lo:hi].view(np.uint32), want.view(np.uint32))
        # a twin of the first planted entry behind everything: equal score, the lower index keeps the result (T.m:80)
        corpus.append_fingerprint(lb.Fingerprint.from_bools(oracle.synth_entry(CSEED, planted[0], int(counts[planted[0]]), 200)))
        again = corpus.query_batch(fps)
        assert [(a[0], _bits(a[1])) for a in again] == [(g[0], _bits(g[1])) for g in got]


@pytest.mark.parametrize("nq", [21, 200, 2000])
def test_bound_pruning_against_adversarial_layouts(lb, gpu, oracle, nq):
    """The bound that lets a top-1 scan give up groups of offsets (k_sliding.hip: kPruneMargin) against the layouts that
    could break it: a strong match is published early, and LATER entries trail it by a hair for most of the query and then
    collect ratios of exactly 1.0 to the end -- one ends in an exact TIE (the lower index must keep the result, strict '<' of
    LBAudioDetectiveTests.m:80), one strictly BETTER by a single sign pair (it must win), others a little inside and a little
    outside the 0.1 % margin.  Pruned scan == full scan == oracle, at queries of 21, 200 and 2000 sub-fingerprints, with the
    threshold at its default and moved."""
    L = 200
    rng = np.random.default_rng(900 + nq)
    q = _rand_fp(rng, nq, L, p_zero=0.0)                      # every pair set: possible = 100 in every sub-fingerprint

    def variant(flips):
        """an entry of nq + 9 sub-fingerprints that holds the query at offset 4 with flips[j] sign pairs of sub-fingerprint j reversed"""
        e = _rand_fp(rng, nq + 9, L, p_zero=0.0)
        w = q.copy()
        for j, k in flips.items():
            for p in range(k):
                w[j, 2 * p], w[j, 2 * p + 1] = w[j, 2 * p + 1], w[j, 2 * p]
        e[4:4 + nq] = w
        return e

    lead = {0: 9, 1: 17, 2: 4, 3: 11}                         # mismatches in the FIRST steps only: the rest of the query scores 1.0
    published = variant(lead)
    tie = variant(lead)                                        # the same ratios in the same order: the same float32 sum
    better = variant({0: 9, 1: 17, 2: 4, 3: 10})              # one pair fewer
    inside = variant({0: 9, 1: 17, 2: 4, 3: 11, 4: max(1, nq // 1250)})                  # trails by < 0.1 %
    outside = variant({0: 30, 1: 30, 2: 30, 3: 30, 4: 30})                             # trails by more
    noise = [_rand_fp(rng, int(n), L, p_zero=0.02) for n in rng.integers(20, 71, 300)]
    for layout, winner in (([published] + noise[:100] + [tie] + noise[100:], "published"),
                           ([tie] + noise[:50] + [published] + noise[50:] + [inside, outside], "first of the tie"),
                           (noise[:10] + [published] + noise[10:200] + [inside, tie, outside] + noise[200:] + [better], "better"),
                           ([outside, inside] + noise[:150] + [better, published, tie] + noise[150:], "better")):
        corpus, _ = _ragged_corpus(lb, gpu, layout, L)
        fq = lb.Fingerprint.from_bools(q)
        bi, bs = oracle.corpus_best_ragged(q, layout, L, nthreads=8)
        assert bs > 0.9
        corpus.set_bound_pruning(False)
        full = corpus.query(fq)
        results = []
        for thr in (0.7, 0.5, 0.95):
            corpus.set_bound_pruning(True).set_bound_pruning_threshold(thr)
            assert abs(corpus.bound_pruning_threshold - thr) < 1e-6
            for _ in range(3):                                 # (which wave publishes first is a race: the result must not be)
                results.append(corpus.query(fq))
        for r in [full] + results:
            assert (r[0], _bits(r[1])) == (bi, _bits(bs)), (nq, winner, r, bi, bs)
        corpus.dispose()
    with pytest.raises(lb.LBAudioDetectiveError):
        corpus2, _ = _ragged_corpus(lb, gpu, noise[:3], L)
        corpus2.set_bound_pruning_threshold(0.0)


def test_pass_of_a_scan_never_spans_more_than_its_offsets_can_address(lb, gpu, oracle):
    """Round-4 advice: the lanes of an "A" pass address their records as a wave-uniform base + a 32-bit byte offset, and in
    a corpus of more than 4 GiB a wave's left-over tasks and its next claim can lie 2^27 records apart (entries without
    tasks of that kind in between).  The corpus here has exactly that shape: a few hundred entries longer than the query,
    then 2^27 + records in 6.9 M entries NOT longer than it, then a few hundred longer ones with the match planted in the last.
    Whichever wave holds tasks of both sides, the pass is cut at the gap (k_sliding.hip: kPassSpan) -- ten queries, every
    one must find the planted entry with the oracle's score."""
    L, nq = 200, 21
    rng = np.random.default_rng(31)
    n_a = 300
    a_entries = [_rand_fp(rng, 30, L, p_zero=0.02) for _ in range(2 * n_a)]          # 30 > 21: three tasks of four offsets each
    q = _rand_fp(rng, nq, L, p_zero=0.0)
    a_entries[-1][5:5 + nq] = q
    a_entries[-1][7, :40] ^= 1
    gap_entries = (1 << 27) // 20 + 1000
    counts = np.concatenate([np.full(n_a, 30, np.uint32), np.full(gap_entries, 20, np.uint32), np.full(n_a, 30, np.uint32)])
    total = int(counts.sum(dtype=np.uint64))
    assert total * 32 > (1 << 32)
    packed = gpu.zeros((total, 32), dtype=gpu.uint8, device="cuda")                  # the gap: all-zero records (they score 0)
    def pack(es):
        return np.stack([lb.pack_subfingerprint(r) for e in es for r in e]).view(np.uint8).reshape(-1, 32)
    packed[: n_a * 30] = gpu.from_numpy(pack(a_entries[:n_a])).cuda()
    packed[total - n_a * 30:] = gpu.from_numpy(pack(a_entries[n_a:])).cuda()
    corpus = lb.Corpus.ragged(L, len(counts), total)
    corpus.append_ragged_packed_device(packed, counts)
    del packed
    bi, bs = oracle.corpus_best_ragged(q, a_entries, L)
    assert bi == 2 * n_a - 1 and bs > 0.9
    want_index = len(counts) - 1
    fq = lb.Fingerprint.from_bools(q)
    for pruning in (False, True):
        corpus.set_bound_pruning(pruning)
        for _ in range(5):
            idx, score = corpus.query(fq)
            assert (idx, _bits(score)) == (want_index, _bits(bs)), (pruning, idx, score, want_index, bs)
    corpus.dispose()
