"""ctypes front end for the CPU parity oracle (oracle/lbad_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (lbaudiodetective_amd) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# LBAD_ORACLE_LIB: another build of the SAME two sources (the `asan` target of oracle/Makefile); never a different oracle
_LIB_PATH = os.path.abspath(os.environ["LBAD_ORACLE_LIB"]) if os.environ.get("LBAD_ORACLE_LIB") else \
    os.path.join(_HERE, "_build", "liblbad_oracle.so")

ROWS_PER_FRAME = 128


class Config(C.Structure):
    _fields_ = [
        ("sample_rate", C.c_double),
        ("window", C.c_uint32),
        ("stride", C.c_uint32),
        ("bands", C.c_uint32),
        ("subfp_len", C.c_uint32),
    ]

    def __init__(self, sample_rate=5512.0, window=2048, stride=64, bands=32, subfp_len=200):
        super().__init__(float(sample_rate), int(window), int(stride), int(bands), int(subfp_len))


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (no GPU involved)."""
    srcs = [os.path.join(_HERE, n) for n in ("lbad_oracle.c", "lbad_file_oracle.c", "lbad_oracle.h", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(os.path.getmtime(p) > os.path.getmtime(_LIB_PATH) for p in srcs)
    if force or stale:
        target = ["asan"] if os.path.basename(_LIB_PATH) == "liblbad_oracle_asan.so" else []
        subprocess.check_call(["make", "-C", _HERE, "--no-print-directory"] + target + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
    u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
    u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
    i16p = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")
    cfgp = C.POINTER(Config)

    def sig(name, res, args):
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args

    sig("lbo_twiddles", None, [C.c_uint32, f32p, f32p])
    sig("lbo_rfft_packed", C.c_int, [f32p, C.c_uint32, f32p])
    sig("lbo_band_table", None, [C.c_double, C.c_uint32, C.c_uint32, C.c_uint32, u32p, u32p, u32p])
    sig("lbo_band_energies", None, [f32p, C.c_uint32, C.c_uint32, u32p, u32p, u32p, f32p])
    sig("lbo_window_row", C.c_int, [f32p, cfgp, f32p])
    sig("lbo_haar_1d", None, [f32p, C.c_uint32])
    sig("lbo_haar_2d", None, [f32p, C.c_uint32, C.c_uint32])
    sig("lbo_extract", None, [f32p, C.c_uint32, C.c_uint32, C.c_uint32, u8p])
    sig("lbo_subfingerprint_count", C.c_uint64, [C.c_uint64, C.c_uint32, C.c_uint32])
    sig("lbo_fingerprint_pcm", C.c_uint64, [f32p, C.c_uint64, cfgp, u8p])
    sig("lbo_fingerprint_pcm_taps", C.c_uint64, [f32p, C.c_uint64, cfgp, u8p, C.c_void_p, C.c_void_p])
    sig("lbo_fingerprint_file_loop", C.c_uint64, [f32p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, cfgp,
                                                   u8p, C.c_void_p, C.c_void_p])
    sig("lbo_spectra_to_rows", C.c_int, [f32p, C.c_uint64, cfgp, f32p])
    sig("lbo_rows_to_subfingerprints", C.c_int, [f32p, C.c_uint64, cfgp, u8p])
    sig("lbo_rfft_packed_batch", C.c_int, [f32p, C.c_uint64, C.c_uint32, f32p])
    sig("lbo_fingerprint_batch", C.c_int, [f32p, C.c_uint64, C.c_uint64, cfgp, u8p, C.c_int])
    sig("lbo_compare_sub", C.c_float, [u8p, u8p, C.c_uint32, C.c_uint32])
    sig("lbo_compare_fp", C.c_float, [u8p, C.c_uint32, u8p, C.c_uint32, C.c_uint32, C.c_uint32])
    sig("lbo_corpus_best", None, [u8p, C.c_uint32, u8p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                  C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_float)])
    sig("lbo_corpus_best_ragged", None, [u8p, C.c_uint32, u8p, u32p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int,
                                         C.POINTER(C.c_int64), C.POINTER(C.c_float), C.c_void_p])
    u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
    sig("lbo_pack_bools", None, [u8p, C.c_uint64, C.c_uint32, u64p])
    sig("lbo_corpus_best_packed", None, [u64p, C.c_uint32, u64p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_float)])
    sig("lbo_synth_ragged_count", C.c_uint32, [C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32])
    sig("lbo_file_decode", C.c_int, [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint64), C.POINTER(C.c_double)])
    sig("lbo_file_decode_bytes", C.c_int, [C.c_char_p, C.c_uint64, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint64),
                                           C.POINTER(C.c_double)])
    sig("lbo_file_free", None, [C.POINTER(C.c_float)])
    sig("lbo_file_set_ima4_carry", None, [C.c_int])
    sig("lbo_resample_count", C.c_uint64, [C.c_uint64, C.c_double, C.c_double])
    sig("lbo_resample", C.c_int, [f32p, C.c_uint64, C.c_double, C.c_double, C.c_int, f32p])
    sig("lbo_fingerprint_file", C.c_int, [C.c_char_p, cfgp, C.c_int, C.c_int, C.c_int, C.POINTER(C.POINTER(C.c_uint8)),
                                          C.POINTER(C.c_uint64)])
    sig("lbo_fingerprint_files", C.c_int, [C.POINTER(C.c_char_p), C.c_uint64, cfgp, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint64)])
    sig("lbo_synth_sine_table", None, [i16p])
    sig("lbo_synth_clip", None, [C.c_uint32, C.c_uint64, C.c_double, C.c_uint32, C.c_int, f32p])
    sig("lbo_synth_entry", None, [C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, u8p])
    _lib = L
    return L


# ---------------------------------------------------------------------------------------------
# thin numpy wrappers
# ---------------------------------------------------------------------------------------------
def twiddles(W: int):
    re = np.empty(W // 2, np.float32)
    im = np.empty(W // 2, np.float32)
    lib().lbo_twiddles(W, re, im)
    return re, im


def rfft_packed(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    if lib().lbo_rfft_packed(x, x.size, out) != 0:
        raise ValueError("window must be a power of two >= 8")
    return out


def band_table(sample_rate: float, window: int, bands: int = 32, n_frames: int | None = None):
    n_frames = window if n_frames is None else n_frames
    idx = np.empty(bands + 1, np.uint32)
    lo = np.empty(bands, np.uint32)
    hi = np.empty(bands, np.uint32)
    lib().lbo_band_table(sample_rate, window, n_frames, bands, idx, lo, hi)
    return idx, lo, hi


def band_energies(spectrum: np.ndarray, idx, lo, hi) -> np.ndarray:
    spectrum = np.ascontiguousarray(spectrum, np.float32)
    out = np.empty(len(lo), np.float32)
    lib().lbo_band_energies(spectrum, spectrum.size, len(lo), idx, lo, hi, out)
    return out


def window_row(pcm_window: np.ndarray, cfg: Config) -> np.ndarray:
    w = np.ascontiguousarray(pcm_window, np.float32)
    out = np.empty(cfg.bands, np.float32)
    if lib().lbo_window_row(w, C.byref(cfg), out) != 0:
        raise ValueError("invalid config")
    return out


def haar_1d(a: np.ndarray) -> np.ndarray:
    a = np.array(a, np.float32, order="C")
    lib().lbo_haar_1d(a, a.size)
    return a


def haar_2d(m: np.ndarray) -> np.ndarray:
    m = np.array(m, np.float32, order="C")
    lib().lbo_haar_2d(m, m.shape[0], m.shape[1])
    return m


def extract(m: np.ndarray, n_wavelets: int) -> np.ndarray:
    m = np.ascontiguousarray(m, np.float32)
    out = np.zeros(2 * n_wavelets, np.uint8)
    lib().lbo_extract(m, m.shape[0], m.shape[1], n_wavelets, out)
    return out


def subfingerprint_count(n_samples: int, window: int, stride: int) -> int:
    return int(lib().lbo_subfingerprint_count(n_samples, window, stride))


def fingerprint_pcm(pcm: np.ndarray, cfg: Config, taps: bool = False):
    pcm = np.ascontiguousarray(pcm, np.float32)
    n = subfingerprint_count(pcm.size, cfg.window, cfg.stride)
    out = np.zeros((n, cfg.subfp_len), np.uint8)
    if not taps:
        got = lib().lbo_fingerprint_pcm(pcm, pcm.size, C.byref(cfg), out.reshape(-1) if n else np.zeros(1, np.uint8))
        if got == 2**64 - 1:
            raise ValueError("invalid config")
        return out
    raw = np.zeros((n, ROWS_PER_FRAME, cfg.bands), np.float32)
    haar = np.zeros_like(raw)
    got = lib().lbo_fingerprint_pcm_taps(
        pcm, pcm.size, C.byref(cfg), out.reshape(-1) if n else np.zeros(1, np.uint8),
        raw.ctypes.data_as(C.c_void_p), haar.ctypes.data_as(C.c_void_p))
    if got == 2**64 - 1:
        raise ValueError("invalid config")
    return out, raw, haar


TAIL_ZERO_FILL, TAIL_NOTHING, TAIL_STALE = 0, 1, 2


def fingerprint_file_loop(client: np.ndarray, file_frames: int, hop: int, cfg: Config, tail_mode: int = TAIL_NOTHING,
                          taps: bool = False):
    """Upstream's file loop with file-frame bookkeeping and short reads (lbo_fingerprint_file_loop)."""
    client = np.ascontiguousarray(client, np.float32)
    n = 0
    if file_frames >= cfg.window:
        n = ((file_frames - cfg.window) // cfg.stride) // ROWS_PER_FRAME
    out = np.zeros((n, cfg.subfp_len), np.uint8)
    raw = np.zeros((n, ROWS_PER_FRAME, cfg.bands), np.float32)
    n_read = np.zeros(n * ROWS_PER_FRAME, np.uint32)
    got = lib().lbo_fingerprint_file_loop(
        client if client.size else np.zeros(1, np.float32), client.size, file_frames, hop, tail_mode, C.byref(cfg),
        out.reshape(-1) if n else np.zeros(1, np.uint8),
        raw.ctypes.data_as(C.c_void_p) if n else None, n_read.ctypes.data_as(C.c_void_p) if n else None)
    if got == 2**64 - 1:
        raise ValueError("invalid config")
    return (out, raw, n_read) if taps else out


def rfft_packed_batch(windows: np.ndarray) -> np.ndarray:
    """Canonical FFT of every row of a [n, W] float32 array."""
    windows = np.ascontiguousarray(windows, np.float32)
    out = np.empty_like(windows)
    if lib().lbo_rfft_packed_batch(windows.reshape(-1), windows.shape[0], windows.shape[1], out.reshape(-1)) != 0:
        raise ValueError("window must be a power of two >= 8")
    return out


def spectra_to_rows(spectra: np.ndarray, cfg: Config) -> np.ndarray:
    """Band rows (LBAudioDetective.m:361-405) of [n, W] packed spectra from any FFT."""
    spectra = np.ascontiguousarray(spectra, np.float32)
    rows = np.empty((spectra.shape[0], cfg.bands), np.float32)
    if lib().lbo_spectra_to_rows(spectra.reshape(-1), spectra.shape[0], C.byref(cfg), rows.reshape(-1)) != 0:
        raise ValueError("invalid config")
    return rows


def rows_to_subfingerprints(rows: np.ndarray, cfg: Config) -> np.ndarray:
    """Haar + ranked signs + truncation for [n_frames * 128, bands] rows."""
    rows = np.ascontiguousarray(rows, np.float32)
    n = rows.shape[0] // ROWS_PER_FRAME
    out = np.zeros((n, cfg.subfp_len), np.uint8)
    if n and lib().lbo_rows_to_subfingerprints(rows.reshape(-1), n, C.byref(cfg), out.reshape(-1)) != 0:
        raise ValueError("invalid config")
    return out


def fingerprint_batch(pcm: np.ndarray, cfg: Config, nthreads: int = 1) -> np.ndarray:
    pcm = np.ascontiguousarray(pcm, np.float32)
    n_clips, spc = pcm.shape
    per = subfingerprint_count(spc, cfg.window, cfg.stride)
    out = np.zeros((n_clips, per, cfg.subfp_len), np.uint8)
    rc = lib().lbo_fingerprint_batch(pcm.reshape(-1), n_clips, spc, C.byref(cfg),
                                     out.reshape(-1) if out.size else np.zeros(1, np.uint8), nthreads)
    if rc != 0:
        raise ValueError("invalid config")
    return out


def compare_sub(a: np.ndarray, b: np.ndarray, range_: int) -> float:
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return float(lib().lbo_compare_sub(a, b, a.size, range_))


def compare_fp(fp1: np.ndarray, fp2: np.ndarray, range_: int, subfp_len: int | None = None) -> float:
    fp1 = np.ascontiguousarray(fp1, np.uint8)
    fp2 = np.ascontiguousarray(fp2, np.uint8)
    L = subfp_len if subfp_len is not None else (fp1.shape[1] if fp1.ndim == 2 and fp1.shape[0] else fp2.shape[1])
    n1 = fp1.shape[0] if fp1.ndim == 2 else 0
    n2 = fp2.shape[0] if fp2.ndim == 2 else 0
    d1 = fp1.reshape(-1) if fp1.size else np.zeros(1, np.uint8)
    d2 = fp2.reshape(-1) if fp2.size else np.zeros(1, np.uint8)
    return float(lib().lbo_compare_fp(d1, n1, d2, n2, L, range_))


def corpus_best(query: np.ndarray, corpus: np.ndarray, range_: int, nthreads: int = 1):
    query = np.ascontiguousarray(query, np.uint8)
    corpus = np.ascontiguousarray(corpus, np.uint8)
    n_entries, n_sub, L = corpus.shape
    bi = C.c_int64(-1)
    bs = C.c_float(0.0)
    lib().lbo_corpus_best(query.reshape(-1), query.shape[0], corpus.reshape(-1), n_entries, n_sub, L, range_,
                          nthreads, C.byref(bi), C.byref(bs))
    return int(bi.value), float(bs.value)


def pack_bools(bools: np.ndarray) -> np.ndarray:
    """[..., L] Booleans (L <= 256) -> [..., 4] uint64 words, Boolean b = bit b & 63 of word b >> 6."""
    bools = np.ascontiguousarray(bools, np.uint8)
    L = bools.shape[-1]
    rows = bools.size // L if L else 0
    out = np.zeros((max(rows, 1), 4), np.uint64)
    if rows:
        lib().lbo_pack_bools(bools.reshape(-1), rows, L, out.reshape(-1))
    return out[:rows].reshape(bools.shape[:-1] + (4,))


def corpus_best_packed(query_words: np.ndarray, corpus_words: np.ndarray, subfp_len: int, range_: int, nthreads: int = 1):
    """lbo_corpus_best on packed rows (pack_bools): query [n_query, 4], corpus [n_entries, n_sub, 4] uint64."""
    query_words = np.ascontiguousarray(query_words, np.uint64)
    corpus_words = np.ascontiguousarray(corpus_words, np.uint64)
    n_entries, n_sub, _ = corpus_words.shape
    bi, bs = C.c_int64(-1), C.c_float(0.0)
    lib().lbo_corpus_best_packed(query_words.reshape(-1), query_words.shape[0],
                                 corpus_words.reshape(-1) if corpus_words.size else np.zeros(4, np.uint64),
                                 n_entries, n_sub, subfp_len, range_, nthreads, C.byref(bi), C.byref(bs))
    return int(bi.value), float(bs.value)


def corpus_best_ragged(query: np.ndarray, entries, range_: int, nthreads: int = 1, want_scores: bool = False):
    """Best-match loop over entries of different lengths: `entries` is a list of [n_e, L] Boolean arrays (or a
    tuple (flat [sum n_e, L] array, counts))."""
    query = np.ascontiguousarray(query, np.uint8)
    if isinstance(entries, tuple):
        flat, counts = entries
        flat = np.ascontiguousarray(flat, np.uint8)
        counts = np.ascontiguousarray(counts, np.uint32)
    else:
        counts = np.array([e.shape[0] for e in entries], np.uint32)
        flat = np.ascontiguousarray(np.concatenate([np.asarray(e, np.uint8) for e in entries], axis=0))
    L = query.shape[1]
    bi, bs = C.c_int64(-1), C.c_float(0.0)
    scores = np.zeros(len(counts), np.float32) if want_scores else None
    lib().lbo_corpus_best_ragged(query.reshape(-1), query.shape[0], flat.reshape(-1) if flat.size else np.zeros(1, np.uint8),
                                 counts if counts.size else np.zeros(1, np.uint32), len(counts), L, range_, nthreads,
                                 C.byref(bi), C.byref(bs), scores.ctypes.data_as(C.c_void_p) if want_scores else None)
    return (int(bi.value), float(bs.value), scores) if want_scores else (int(bi.value), float(bs.value))


def synth_ragged_counts(seed: int, first: int, count: int, lo: int, hi: int) -> np.ndarray:
    """Sub-fingerprint counts of entries first .. first + count - 1 of the synthetic ragged corpus (vectorised
    restatement of lbo_synth_ragged_count; checked against it in tests/test_oracle.py)."""
    def mix32(x):
        x = x.astype(np.uint32)
        x ^= x >> np.uint32(16); x *= np.uint32(0x7feb352d)
        x ^= x >> np.uint32(15); x *= np.uint32(0x846ca68b)
        x ^= x >> np.uint32(16)
        return x
    e = np.arange(first, first + count, dtype=np.uint64)
    lo32 = (e & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi32 = (e >> np.uint64(32)).astype(np.uint32)
    with np.errstate(over="ignore"):
        r = mix32(np.uint32(seed & 0xFFFFFFFF) ^ np.uint32(0x52414747) ^ mix32(lo32) ^ (hi32 * np.uint32(0x632BE5AB)))
    return (np.uint32(lo) + r % np.uint32(hi - lo + 1)).astype(np.uint32)


def synth_ragged_entries(seed: int, first: int, counts: np.ndarray, subfp_len: int) -> np.ndarray:
    """Booleans of the entries first .. first + len(counts) - 1, back to back: [sum counts, subfp_len]."""
    out = np.zeros((int(counts.sum()), subfp_len), np.uint8)
    at = 0
    for i, n in enumerate(counts):
        lib().lbo_synth_entry(seed & 0xFFFFFFFF, first + i, int(n), subfp_len, out[at:at + int(n)].reshape(-1))
        at += int(n)
    return out


# ---- file front end (lbad_file_oracle.c) -------------------------------------------------------------------
def decode_audio_file(path: str = None, data: bytes = None):
    """A CAF / WAV file (or its bytes) -> (mono float32 at the file's rate, rate); ValueError if unsupported."""
    buf, n, rate = C.POINTER(C.c_float)(), C.c_uint64(0), C.c_double(0.0)
    if data is not None:
        rc = lib().lbo_file_decode_bytes(data, len(data), C.byref(buf), C.byref(n), C.byref(rate))
    else:
        rc = lib().lbo_file_decode(path.encode(), C.byref(buf), C.byref(n), C.byref(rate))
    if rc == -43:
        raise FileNotFoundError(path)
    if rc != 0:
        raise ValueError(f"unsupported audio file (oracle status {rc})")
    try:
        out = np.ctypeslib.as_array(buf, shape=(n.value,)).copy() if n.value else np.zeros(0, np.float32)
    finally:
        lib().lbo_file_free(buf)
    return out, float(rate.value)


def resample(x: np.ndarray, rate_in: float, rate_out: float, model: int = 0) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    n = int(lib().lbo_resample_count(x.size, rate_in, rate_out))
    out = np.zeros(max(n, 1), np.float32)
    if lib().lbo_resample(x if x.size else np.zeros(1, np.float32), x.size, rate_in, rate_out, model, out) != 0:
        raise ValueError("unsupported conversion")
    return out[:n]


def fingerprint_file(path: str, cfg: Config, hop_mode: int = 1, tail_mode: int = TAIL_NOTHING, resampler: int = 0) -> np.ndarray:
    """decode + convert + upstream's window loop, all in the oracle: [count, subfp_len] Booleans."""
    buf, n = C.POINTER(C.c_uint8)(), C.c_uint64(0)
    rc = lib().lbo_fingerprint_file(path.encode(), C.byref(cfg), hop_mode, tail_mode, resampler, C.byref(buf), C.byref(n))
    if rc == -43:
        raise FileNotFoundError(path)
    if rc != 0:
        raise ValueError(f"oracle status {rc}")
    try:
        out = np.ctypeslib.as_array(buf, shape=(n.value * cfg.subfp_len,)).copy().reshape(n.value, cfg.subfp_len) \
            if n.value else np.zeros((0, cfg.subfp_len), np.uint8)
    finally:
        C.CDLL(None).free(buf)
    return out


def fingerprint_files(paths, cfg: Config, hop_mode: int = 1, tail_mode: int = TAIL_NOTHING, resampler: int = 0, nthreads: int = 1):
    """fingerprint_file over many files, one file per OpenMP thread (lbo_fingerprint_files): list of [count, subfp_len] arrays."""
    n = len(paths)
    arr = (C.c_char_p * n)(*[p.encode() for p in paths])
    bufs = (C.POINTER(C.c_uint8) * n)()
    counts = (C.c_uint64 * n)()
    rc = lib().lbo_fingerprint_files(arr, n, C.byref(cfg), hop_mode, tail_mode, resampler, nthreads, bufs, counts)
    out = []
    try:
        if rc != 0:
            raise ValueError(f"oracle status {rc}")
        for i in range(n):
            k = int(counts[i])
            out.append(np.ctypeslib.as_array(bufs[i], shape=(k * cfg.subfp_len,)).copy().reshape(k, cfg.subfp_len)
                       if k else np.zeros((0, cfg.subfp_len), np.uint8))
    finally:
        for i in range(n):
            if bufs[i]:
                C.CDLL(None).free(bufs[i])
    return out


def synth_sine_table() -> np.ndarray:
    t = np.zeros(1024, np.int16)
    lib().lbo_synth_sine_table(t)
    return t


def synth_clip(seed: int, clip: int, sample_rate: float, n_samples: int, stereo_sum: bool = False) -> np.ndarray:
    out = np.empty(n_samples, np.float32)
    lib().lbo_synth_clip(seed & 0xFFFFFFFF, clip, sample_rate, n_samples, int(stereo_sum), out)
    return out


def synth_clips(seed: int, first: int, count: int, sample_rate: float, n_samples: int, stereo_sum=False):
    return np.stack([synth_clip(seed, first + i, sample_rate, n_samples, stereo_sum) for i in range(count)])


def synth_entry(seed: int, entry: int, n_sub: int, subfp_len: int) -> np.ndarray:
    out = np.zeros((n_sub, subfp_len), np.uint8)
    lib().lbo_synth_entry(seed & 0xFFFFFFFF, entry, n_sub, subfp_len, out.reshape(-1))
    return out


def synth_corpus(seed: int, first: int, count: int, n_sub: int, subfp_len: int) -> np.ndarray:
    return np.stack([synth_entry(seed, first + i, n_sub, subfp_len) for i in range(count)])
