"""CPU tests of the parity oracle: pin it to the reference's known answers and to an
independent restatement, so the GPU parity tests compare against something trustworthy."""
import json
import os

import numpy as np
import pytest

SEED = 0x4C424144

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_haar_known_answer(oracle):
    """Upstream testHaarWaveletDecomposition (LBAudioDetectiveTests.m:157-176) + essay Fig. 13."""
    g = json.load(open(os.path.join(GOLD, "haar_known_answer.json")))
    got = oracle.haar_2d(np.array(g["input"], np.float32))
    assert np.array_equal(np.rint(got).astype(int), np.array(g["expected_integers"]))
    assert np.allclose(got, np.array(g["expected_float32"], np.float32), rtol=0, atol=6e-5)


def test_haar_1d_non_power_of_two(oracle):
    # length 3: pre-scale all, butterfly only the first pair (integer halving, Frame.m:143-152)
    a = np.array([3.0, 5.0, 7.0], np.float32)
    got = oracle.haar_1d(a)
    r3, r2 = np.sqrt(np.float32(3)), np.sqrt(np.float32(2))
    s = a / r3
    want = np.array([(s[0] + s[1]) / r2, (s[0] - s[1]) / r2, s[2]], np.float32)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("name", ["A", "B", "C"])
def test_band_tables_match_survey(oracle, name):
    g = json.load(open(os.path.join(GOLD, "band_tables.json")))[name]
    idx, lo, hi = oracle.band_table(g["sample_rate"], g["window"])
    assert list(idx) == g["indices"]
    if "ranges" in g:
        assert [[int(a), int(b)] for a, b in zip(lo, hi)] == g["ranges"]
    else:
        assert list(lo) == g["lo"] and list(hi) == g["hi"]
    if "divisors" in g:
        assert list(np.diff(idx.astype(np.int64))) == g["divisors"]


def test_band_quirks_at_44k(oracle):
    """at 44.1 kHz / 1024 only bins 0..21 are read and 17 bands are empty."""
    _, lo, hi = oracle.band_table(44100, 1024)
    assert int(hi.max()) == 22 and int(lo.min()) == 0
    assert int((lo == hi).sum()) == 17
    x = np.zeros(1024, np.float32)
    x[0] = 1.0  # impulse: flat spectrum, 2*DFT = 2 everywhere
    row = oracle.window_row(x, oracle.Config(44100, 1024))
    assert np.count_nonzero(row) == 15


def _rfft_packed_f64(x):
    ref = np.fft.rfft(x.astype(np.float64)) * 2
    out = np.empty(x.size)
    out[0], out[1] = ref[0].real, ref[-1].real
    out[2::2], out[3::2] = ref[1:-1].real, ref[1:-1].imag
    return out


@pytest.mark.parametrize("W", [8, 16, 64, 512, 1024, 2048, 4096, 8192])
def test_fft_against_float64(oracle, W):
    rng = np.random.default_rng(W)
    x = (rng.integers(-32768, 32768, W) / 32768).astype(np.float32)
    got = oracle.rfft_packed(x).astype(np.float64)
    want = _rfft_packed_f64(x)
    # float32 radix-2 error grows ~ log2(W) * eps * |X|
    assert np.abs(got - want).max() <= 4e-7 * np.log2(W) * np.abs(want).max()


def test_fft_packing_dc_and_nyquist(oracle):
    W = 64
    x = np.ones(W, np.float32)
    y = oracle.rfft_packed(x)
    assert y[0] == 2 * W and y[1] == 0 and not y[2:].any()
    x = np.where(np.arange(W) % 2 == 0, 1, -1).astype(np.float32)
    y = oracle.rfft_packed(x)
    assert y[0] == 0 and y[1] == 2 * W and not y[2:].any()


def test_twiddle_symmetry(oracle):
    re, im = oracle.twiddles(1024)
    assert re[0] == 1 and im[0] == 0 and re[256] == 0 and im[256] == -1
    assert re[128] == -im[128]
    assert np.array_equal(re[1:256], -re[512 - 1:256:-1])
    assert np.array_equal(re[1:128], -im[255:128:-1])


def _bands_numpy(spec, idx, lo, hi, n_frames):
    """Independent float32 restatement of LBAudioDetective.m:373-405."""
    norm = np.float32((n_frames // 2) // 2)
    out = np.zeros(len(lo), np.float32)
    for i in range(len(lo)):
        p = np.float32(0)
        for k in range(int(lo[i]), int(hi[i])):
            re, im = spec[2 * k], spec[2 * k + 1]
            if re > 0:
                re = np.float32(re / norm)
            if im > 0:
                im = np.float32(im / norm)
            v = np.float32(np.float32(re * re) + np.float32(im * im))
            if np.isfinite(v):
                p = np.float32(p + v)
        with np.errstate(divide="ignore", invalid="ignore"):
            out[i] = np.float32(p) / np.float32(int(idx[i + 1]) - int(idx[i]))
    return out


@pytest.mark.parametrize("rate,W", [(5512, 2048), (44100, 1024), (48000, 4096)])
def test_band_energies_against_numpy(oracle, rate, W):
    rng = np.random.default_rng(rate)
    spec = (rng.standard_normal(W) * 50).astype(np.float32)
    idx, lo, hi = oracle.band_table(rate, W)
    assert np.array_equal(oracle.band_energies(spec, idx, lo, hi), _bands_numpy(spec, idx, lo, hi, W))


def _extract_numpy(m, n_wavelets):
    flat = m.reshape(-1)
    order = np.argsort(-np.abs(flat.astype(np.float64)), kind="stable")  # ties: ascending index
    out = np.zeros(2 * n_wavelets, np.uint8)
    for i in range(min(n_wavelets, flat.size)):
        v = flat[order[i]]
        if v > 0:
            out[2 * i] = 1
        elif v < 0:
            out[2 * i + 1] = 1
    return out


def test_extract_against_numpy_with_ties(oracle):
    rng = np.random.default_rng(5)
    m = rng.standard_normal((128, 32)).astype(np.float32)
    m[rng.random(m.shape) < 0.3] = 0.0
    m[5, 5] = m[9, 1] = 3.25       # equal magnitudes, different positions
    m[7, 7] = -3.25
    assert np.array_equal(oracle.extract(m, 200), _extract_numpy(m, 200))
    assert np.array_equal(oracle.extract(np.zeros((128, 32), np.float32), 200), np.zeros(400, np.uint8))


def _compare_sub_py(a, b, length, rng_):
    poss = hits = 0
    for i in range(0, min(rng_, length), 2):
        a0, a1 = a[i], a[i + 1] if i + 1 < length else 0
        if a0 or a1:
            poss += 1
            b0, b1 = b[i], b[i + 1] if i + 1 < length else 0
            if a0 == b0 and a1 == b1:
                hits += 1
    return np.float32(0) if poss == 0 else np.float32(np.float32(hits) / np.float32(poss))


def _compare_fp_py(f1, f2, rng_):
    """Independent restatement of Fingerprint.m:119-149."""
    if len(f1) < len(f2):
        f1, f2 = f2, f1
    n1, n2, L = len(f1), len(f2), f1.shape[1]
    match = np.float32(0)
    for off in range(n1 - n2 + 1):
        s = np.float32(0)
        for i in range(n2):
            s = np.float32(s + _compare_sub_py(f1[i + off], f2[i], L, rng_))
        cand = np.float32(s / np.float32(n2))
        match = cand if match < cand else match
    return match


def test_compare_against_python(oracle):
    g = np.load(os.path.join(GOLD, "oracle_vectors.npz"))
    pa = pb = 0
    for (n1, n2, rg), bits in zip(g["cmp_cases"], g["cmp_expected_bits"]):
        a = g["cmp_a"][pa:pa + n1 * 200].reshape(n1, 200); pa += n1 * 200
        b = g["cmp_b"][pb:pb + n2 * 200].reshape(n2, 200); pb += n2 * 200
        want = _compare_fp_py(a, b, int(rg))
        got = np.float32(oracle.compare_fp(a, b, int(rg)))
        assert got.view(np.uint32) == want.view(np.uint32) == bits, (n1, n2, rg)


def test_compare_invariants(oracle):
    """What the upstream tests rely on: a copy compares equal and scores 1.0
    (LBAudioDetectiveTests.m:119-155); unrelated fingerprints sit near 0.5 (essay p.43)."""
    a = oracle.synth_corpus(1, 0, 1, 5, 200)[0]
    b = oracle.synth_corpus(1, 1, 1, 5, 200)[0]
    assert oracle.compare_fp(a, a.copy(), 200) == 1.0
    assert 0.4 < oracle.compare_fp(a, b, 200) < 0.6
    zero = np.zeros_like(a)
    assert oracle.compare_fp(zero, a, 200) == 0.0          # no possible hits -> 0 (Fp.m:171-173)
    assert oracle.compare_fp(a[:0], a, 200, subfp_len=200) == 0.0   # empty side: NaN never wins Foundation's MAX


def test_packed_popcount_scan_equals_boolean_loop(oracle):
    """The packed popcount CPU scan (bench.py's second compare baseline) returns what the Boolean loop
    restated from Fp.m:119-176 returns -- index and float bits -- for equal and different counts, odd lengths, ranges
    below / above / equal to the length, range 0, ties and all-zero queries."""
    rng = np.random.default_rng(23)
    for nq, ns, n, L, rg in ((5, 5, 3000, 200, 200), (5, 5, 400, 200, 0), (3, 7, 400, 200, 150), (9, 4, 300, 199, 200),
                             (5, 5, 300, 200, 77), (1, 1, 50, 6, 6), (5, 5, 100, 256, 300), (2, 2, 64, 200, 1)):
        corpus = oracle.synth_corpus(7, 0, n, ns, L) if L == 200 else (rng.random((n, ns, L)) < 0.4).astype(np.uint8)
        q = (rng.random((nq, L)) < 0.4).astype(np.uint8)
        if nq == ns:
            q = corpus[n // 2].copy()
            q[0, :10] ^= 1
            corpus[n // 3] = corpus[n // 2]                                     # a tie: the lower index must win
        want = oracle.corpus_best(q, corpus, rg)
        got = oracle.corpus_best_packed(oracle.pack_bools(q), oracle.pack_bools(corpus), L, rg, nthreads=2)
        assert got[0] == want[0] and np.float32(got[1]).view(np.uint32) == np.float32(want[1]).view(np.uint32), (nq, ns, L, rg)
    z = np.zeros((5, 200), np.uint8)
    c = oracle.synth_corpus(7, 0, 10, 5, 200)
    assert oracle.corpus_best_packed(oracle.pack_bools(z), oracle.pack_bools(c), 200, 200) == oracle.corpus_best(z, c, 200) == (-1, 0.0)


def test_ragged_best_match_against_python(oracle):
    """lbo_corpus_best_ragged (entries of different lengths, the shape of LBAudioDetectiveTests.m:57-91) against the
    pure-Python restatement of Fp.m:119-176 entry by entry; strict '<' from 0.0 (T.m:60,80): the lowest index wins
    ties, nothing is selected when every entry scores 0.  And the synthetic ragged corpus's vectorised counts."""
    rng = np.random.default_rng(17)
    def fp(n):
        f = np.zeros((n, 200), np.uint8)
        pos = rng.random((n, 100)) < 0.5
        zero = rng.random((n, 100)) < 0.05
        f[:, 0::2] = pos & ~zero
        f[:, 1::2] = ~pos & ~zero
        return f
    entries = [fp(int(n)) for n in rng.integers(1, 12, 40)]
    entries[7] = entries[3].copy()                                   # a tie between entries 3 and 7
    for nq, rg in ((1, 200), (4, 200), (6, 31), (11, 200), (15, 7)):
        q = fp(nq)
        k = min(nq, entries[3].shape[0])
        q[:k] = entries[3][:k]
        bi, bs, scores = oracle.corpus_best_ragged(q, entries, rg, want_scores=True)
        want = np.array([_compare_fp_py(q, e, rg) for e in entries], np.float32)
        assert np.array_equal(scores.view(np.uint32), want.view(np.uint32)), (nq, rg)
        assert bi == int(np.argmax(want)) and np.float32(bs) == want.max() and bi != 7
    assert oracle.corpus_best_ragged(np.zeros((2, 200), np.uint8), entries, 200) == (-1, 0.0)
    counts = oracle.synth_ragged_counts(0x4C424145, 123_456, 5000, 20, 70)
    assert counts.min() >= 20 and counts.max() <= 70 and len(set(counts.tolist())) == 51
    assert all(int(counts[i]) == oracle.lib().lbo_synth_ragged_count(0x4C424145, 123_456 + i, 20, 70) for i in range(0, 5000, 97))
    ent = oracle.synth_ragged_entries(0x4C424145, 9, counts[:3], 200)
    assert np.array_equal(ent[:int(counts[0])], oracle.synth_entry(0x4C424145, 9, int(counts[0]), 200))


def test_framing_counts(oracle):
    # 1 s at 44.1 kHz / 1024 -> 673 windows -> 5 frames; 48 kHz / 4096 -> 686 -> 5
    assert oracle.subfingerprint_count(44100, 1024, 64) == 5
    assert oracle.subfingerprint_count(48000, 4096, 64) == 5
    assert oracle.subfingerprint_count(1023, 1024, 64) == 0
    assert oracle.subfingerprint_count(1024 + 128 * 64 - 1, 1024, 64) == 0
    assert oracle.subfingerprint_count(1024 + 128 * 64, 1024, 64) == 1


def test_oracle_matches_committed_vectors(oracle):
    g = np.load(os.path.join(GOLD, "oracle_vectors.npz"))
    pcm = g["B_pcm_i16"].astype(np.float32) / np.float32(32768)
    bits, raw, haar = oracle.fingerprint_pcm(pcm[0], oracle.Config(44100, 1024), taps=True)
    assert np.array_equal(bits, g["B_bits"][0])
    assert np.array_equal(raw[0], g["B_raw_frame0"]) and np.array_equal(haar[0], g["B_haar_frame0"])
    pcm = g["A_pcm_i16"].astype(np.float32) / np.float32(32768)
    assert np.array_equal(oracle.fingerprint_pcm(pcm, oracle.Config()), g["A_bits"])
    pcm = g["C_pcm_i32"].astype(np.float32) / np.float32(65536)
    assert np.array_equal(oracle.fingerprint_pcm(pcm, oracle.Config(48000, 4096)), g["C_bits"])


def test_truncation_keeps_top_100_wavelets(oracle):
    """Extract is asked for 200 wavelets but only 200 Booleans (= 100 pairs) survive."""
    g = np.load(os.path.join(GOLD, "oracle_vectors.npz"))
    full = oracle.extract(g["B_haar_frame0"], 200)
    assert np.array_equal(full[:200], g["B_bits"][0][0])
    assert full[:200].reshape(100, 2).sum(axis=1).max() <= 1


def test_synth_is_deterministic_and_int16(oracle):
    a = oracle.synth_clip(0x4C424144, 3, 44100, 2000)
    assert np.array_equal(a, oracle.synth_clip(0x4C424144, 3, 44100, 2000))
    assert np.array_equal(a * 32768, np.rint(a * 32768)) and np.abs(a).max() <= 1
    assert not np.array_equal(a, oracle.synth_clip(0x4C424144, 4, 44100, 2000))
    g = np.load(os.path.join(GOLD, "oracle_vectors.npz"))
    assert np.array_equal(np.round(oracle.synth_clip(0x4C424144, 0, 44100, 44100) * 32768).astype(np.int16), g["B_pcm_i16"][0])


def test_batch_threads_agree(oracle):
    pcm = oracle.synth_clips(9, 0, 6, 44100, 1024 + 128 * 64 * 2)
    cfg = oracle.Config(44100, 1024)
    assert np.array_equal(oracle.fingerprint_batch(pcm, cfg, 1), oracle.fingerprint_batch(pcm, cfg, 4))


# ---------------------------------------------------------------------------------------------
# upstream's file loop (LBAudioDetective.m:236-293) and the essay's figures
# ---------------------------------------------------------------------------------------------
def test_file_loop_tail_modes(oracle):
    cfg = oracle.Config()
    hop, n_client, file_frames = 8, 30000, 30000 * 8 + 5
    pcm = oracle.synth_clip(SEED, 3, 8000, n_client)
    frames = ((file_frames - cfg.window) // cfg.stride) // 128
    first_short = (n_client - cfg.window) // hop + 1
    assert frames * 128 > first_short                      # the tail is inside the windows used
    # zero fill == the plain PCM path on the zero-padded stream with stride = hop
    z, raw_z, nr_z = oracle.fingerprint_file_loop(pcm, file_frames, hop, cfg, oracle.TAIL_ZERO_FILL, taps=True)
    need = frames * 128 * hop + cfg.window
    padded = np.concatenate([pcm, np.zeros(need - n_client, np.float32)])
    assert np.array_equal(z, oracle.fingerprint_pcm(padded, oracle.Config(stride=hop)))
    # nothing read: rows of the short windows are +0.0, the others are untouched
    n, raw_n, nr_n = oracle.fingerprint_file_loop(pcm, file_frames, hop, cfg, oracle.TAIL_NOTHING, taps=True)
    rows_z, rows_n = raw_z.reshape(-1, 32), raw_n.reshape(-1, 32)
    assert np.array_equal(rows_n[:first_short], rows_z[:first_short])
    assert not rows_n[first_short:].any() and (nr_n[first_short:] == 0).all() and (nr_n[:first_short] == cfg.window).all()
    # stale: nRead shrinks by the hop, the first short window still holds mostly samples, and the recursion over
    # the previous spectrum overflows within a few dozen windows (rows of NaN-skipped terms are 0)
    s, raw_s, nr_s = oracle.fingerprint_file_loop(pcm, file_frames, hop, cfg, oracle.TAIL_STALE, taps=True)
    rows_s = raw_s.reshape(-1, 32)
    assert np.array_equal(rows_s[:first_short], rows_z[:first_short])
    assert nr_s[first_short] == n_client - first_short * hop and (np.diff(nr_s[first_short:].astype(np.int64)) == -hop).all()
    assert np.isfinite(rows_s[first_short]).all() and not np.array_equal(rows_s[first_short], rows_z[first_short])
    assert not rows_s[first_short + 100:].any()
    assert z.shape == n.shape == s.shape == (frames, 200)


def test_oracle_reproduces_essay_figures(oracle):
    """The oracle -- its OWN container reader, IMA4 / LPCM decoder and converter (oracle/lbad_file_oracle.c; no product
    code runs), then upstream's file loop -- on upstream's sixty fixtures, against the fifty numbers of the essay's Fig. 24-28
    (tests/golden/essay_figures.json) -- the only end-to-end results the reference publishes.
    Bounds and the two fixtures that cannot be reached: tools/birds_matrix.py."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools"))
    import birds_matrix as bm
    ms = bm.matrices("oracle")
    assert bm.check(ms) == [], bm.summarize(ms)
    d1 = np.diag(ms["test1"])
    lossless = [i for i, b in enumerate(bm.BIRDS) if b not in bm.UNREACHABLE_TEST1]
    assert np.abs(d1 - np.array(bm.ESSAY["tests"]["test1"]["right"]))[lossless].max() < 0.6


def test_fft_rounding_order_hardly_moves_a_fingerprint(oracle):
    """The reference's FFT is Apple's closed-source vDSP (LBAudioDetective.m:353-355); the oracle fixes one
    float32 evaluation order.  tools/vdsp_gap_probe.py runs the rest of the pipeline behind other evaluations of
    the same transform (float64, pocketfft's float32 mixed radix, decimation in frequency without FMA); the full
    run is committed as profiles/r02_vdsp_gap.json (worst variant: 24 of 803 600 bits, no bird fingerprint
    touched).  Here a small sample must stay below 2e-4 flipped bits, with flips only in whole rank swaps."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools"))
    import vdsp_gap_probe as probe
    rep = probe.run("tiny", ["f64", "pocket32", "dif_nofma"])
    for name, r in rep["variants"].items():
        assert r["total"]["bits"] > 30000
        assert r["total"]["flip_rate"] <= 2e-4, (name, r["total"])
        assert r["total"]["flipped"] % 4 == 0, (name, r["total"])      # two sign pairs trade places
        assert r["birds"]["max_match_shift"] <= 0.005, (name, r["birds"])
    full = json.load(open(os.path.join(os.path.dirname(os.path.dirname(__file__)), "profiles", "r02_vdsp_gap.json")))
    assert max(v["total"]["flip_rate"] for v in full["variants"].values()) < 1e-4
    assert all(v["birds"]["subfingerprints_touched"] == 0 for v in full["variants"].values())


def test_files_in_parallel_equal_files_one_by_one():
    """lbo_fingerprint_files (one file per OpenMP thread: bench.py's CPU baseline of the file leg) against lbo_fingerprint_file."""
    import glob
    from oracle import oracle as O
    paths = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "birds", "*.caf")))[:6]
    cfg = O.Config()
    many = O.fingerprint_files(paths, cfg, 1, O.TAIL_NOTHING, 0, 4)
    for p, got in zip(paths[:3], many[:3]):
        assert np.array_equal(got, O.fingerprint_file(p, cfg, 1, O.TAIL_NOTHING, 0))
    assert all(m.shape[1] == cfg.subfp_len for m in many)


def test_oracle_under_sanitizers():
    """SURVEY section 5, "ASan/UBSan build of the CPU oracle": oracle/Makefile's `asan` target (both sources at -O1 with
    AddressSanitizer + UBSan, no recovery) runs this file's known answers, the committed vectors, the tail modes of the
    file loop and upstream's sixty fixtures through oracle/lbad_file_oracle.c (the essay-figure check) in a child
    interpreter with the sanitizer runtime preloaded.  Any report aborts the child.  CPU build only: sanitizers never run
    on the GPU box."""
    import shutil
    import subprocess
    import sys
    if os.environ.get("LBAD_ORACLE_LIB"):
        pytest.skip("already inside the sanitizer run")
    if shutil.which("gcc") is None:
        pytest.skip("no host compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    build = subprocess.run(["make", "-C", os.path.join(root, "oracle"), "--no-print-directory", "asan"], capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr[-2000:]
    runtime = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(runtime) or not os.path.exists(runtime):
        pytest.skip("libasan.so not found")
    env = dict(os.environ)
    env.update({"LBAD_ORACLE_LIB": os.path.join(root, "oracle", "_build", "liblbad_oracle_asan.so"), "LD_PRELOAD": runtime,
                # the interpreter itself is not instrumented: its arenas are not leaks of ours
                "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1:halt_on_error=1", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1",
                "OMP_NUM_THREADS": "4"})
    picked = ("haar or band or fft_against or fft_packing or twiddle or extract or compare or popcount or ragged or framing "
              "or committed or truncation or synth or batch_threads or tail_modes or essay_figures or in_parallel")
    run = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-p", "no:cacheprovider", "-k", picked],
                         capture_output=True, text=True, env=env, cwd=root, timeout=1500)
    assert run.returncode == 0, (run.stdout[-3000:], run.stderr[-3000:])
    assert "AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr, run.stderr[-3000:]
    assert " passed" in run.stdout and "failed" not in run.stdout, run.stdout[-1000:]
