"""CPU tests of the C ABI: the library loads, exports every symbol the header declares, the
host-side containers behave like the upstream ones, and compute entry points refuse to run
(loudly, no fallback) when there is no GPU."""
import ctypes as C
import os

import numpy as np
import pytest


def _has_gpu():
    import torch
    return torch.cuda.is_available()


def test_exports_every_declared_symbol(lb):
    funcs, consts = lb._native.declared_symbols()
    assert len(funcs) >= 60 and len(consts) == 10
    raw = C.CDLL(lb.LIB_PATH)
    for name in funcs + consts:
        assert hasattr(raw, name), f"{name} declared in include/lbaudiodetective.h but not exported"
    assert sorted(lb._native._SIGNATURES) == funcs
    assert sorted(lb._native.CONSTANTS) == consts


def test_constants(lb):
    # LBAudioDetective.m:20-26
    assert lb.constant("kLBAudioDetectiveArgumentInvalid") == 1
    assert lb.constant("kLBAudioDetectiveDefaultWindowSize") == 2048
    assert lb.constant("kLBAudioDetectiveDefaultAnalysisStride") == 64
    assert lb.constant("kLBAudioDetectiveDefaultNumberOfPitchSteps") == 32
    assert lb.constant("kLBAudioDetectiveDefaultSubfingerprintLength") == 200


def test_detective_defaults_and_setters(lb):
    d = lb.Detective()
    assert (d.processing_sample_rate, d.window_size, d.analysis_stride, d.number_of_pitch_steps,
            d.subfingerprint_length) == (5512.0, 2048, 64, 32, 200)
    f = lb.lib().LBAudioDetectiveDefaultProcessingFormat()   # LBAudioDetective.m:116-131
    assert (f.mSampleRate, f.mFormatID, f.mFormatFlags, f.mBitsPerChannel, f.mChannelsPerFrame,
            f.mBytesPerFrame, f.mBytesPerPacket, f.mFramesPerPacket) == (5512.0, 0x6C70636D, 9, 32, 1, 4, 4, 1)
    d.configure(sample_rate=44100, window=1024, stride=32, bands=16, subfp_len=64)
    assert (d.processing_sample_rate, d.window_size, d.analysis_stride, d.number_of_pitch_steps,
            d.subfingerprint_length) == (44100.0, 1024, 32, 16, 64)
    # SetWindowSize: power of two -> noErr, anything else -> ArgumentInvalid and no change
    assert d.set_window_size_status(4096) == 0 and d.window_size == 4096
    assert d.set_window_size_status(1000) == 1 and d.window_size == 4096
    assert d.set_window_size_status(0) == 1 and d.set_window_size_status(1 << 20) == 1
    assert lb.lib().LBAudioDetectiveDispose(None) == 1      # LBAudioDetective.m:93-95
    d.dispose()


def test_subfingerprint_count(lb):
    d = lb.Detective().configure(sample_rate=44100, window=1024)
    assert d.subfingerprint_count(44100) == 5
    assert d.subfingerprint_count(1000) == 0                # shorter than a window
    assert d.subfingerprint_count(1024 + 128 * 64) == 1


def test_fingerprint_container(lb):
    """Container semantics of LBAudioDetectiveFingerprint.m:18-117 (host memory only)."""
    rng = np.random.default_rng(3)
    fp = lb.Fingerprint(0)
    assert fp.subfingerprint_length == 0 and fp.number_of_subfingerprints == 0
    assert fp.set_subfingerprint_length(200) == (True, 200)
    rows = rng.integers(0, 2, (3, 200)).astype(np.uint8)
    for r in rows:
        fp.add_subfingerprint(r)
    assert fp.number_of_subfingerprints == 3
    assert fp.set_subfingerprint_length(100) == (False, 200)    # frozen once non-empty (Fp.m:81-89)
    assert np.array_equal(fp.to_bools(), rows)
    cp = fp.copy()                                              # upstream testFingerprintComparison
    assert fp.equal_to_fingerprint(cp) and cp.equal_to_fingerprint(fp)
    cp.add_subfingerprint(rows[0])
    assert not fp.equal_to_fingerprint(cp)
    assert fp.to_string().count("+") == 2 and len(fp.to_string()) == 3 * 200 + 2
    lb.lib().LBAudioDetectiveFingerprintDispose(None)           # NULL tolerated (Fp.m:29-31)


def test_frame_container(lb):
    """LBAudioDetectiveFrame.m:22-105,155-161,193-210."""
    fr = lb.Frame(3)
    assert not fr.full() and fr.number_of_rows == 0
    assert fr.set_row([1, 2, 3, 4], 0) and fr.set_row([5, 6, 7, 8, 9], 1) and fr.set_row([9, 8, 7, 6], 2)
    assert fr.full() and not fr.set_row([0, 0, 0, 0], 2)
    assert fr.number_of_rows == 3 and fr.fingerprint_length() == 3 * 4 * 2 and fr.fingerprint_size() == 24
    assert fr.get_value(1, 2) == 7.0
    assert np.array_equal(fr.get_row(2, 4), np.array([9, 8, 7, 6], np.float32))
    cp = fr.copy()
    assert fr.equal_to_frame(cp)
    other = lb.Frame(3)
    for i, r in enumerate([[1, 2, 3, 4], [5, 6, 7, 8], [9, 8, 7, 5]]):
        other.set_row(r, i)
    assert not fr.equal_to_frame(other)
    lb.lib().LBAudioDetectiveFrameDispose(None)


def test_pack_unpack_roundtrip(lb):
    rng = np.random.default_rng(0)
    for L in (1, 2, 31, 32, 33, 199, 200, 255, 256):
        b = rng.integers(0, 2, L).astype(np.uint8)
        w = lb.pack_subfingerprint(b)
        assert np.array_equal(lb.unpack_subfingerprint(w, L), b)
        assert np.array_equal(lb.unpack_packed(w[None, :], L)[0], b)
        assert int(sum(bin(int(x)).count("1") for x in w)) == int(b.sum())


def test_decode_key(lb):
    import struct
    bits = struct.unpack("<I", struct.pack("<f", 0.75))[0]
    assert lb.Corpus.decode_key((bits << 32) | (0xFFFFFFFF - 123)) == (123, 0.75)
    assert lb.Corpus.decode_key(0xFFFFFFFF - 5) == (-1, 0.0)     # best score 0 selects nothing (Tests.m:80)
    assert lb.Corpus.decode_key(0) == (-1, 0.0)
    from lbaudiodetective_amd import sharded
    assert sharded.decode_key(sharded.make_key(bits, 123)) == (123, 0.75)


def test_empty_fingerprint_compare_is_zero(lb):
    a, b = lb.Fingerprint(200), lb.Fingerprint(200)
    b.add_subfingerprint(np.ones(200, np.uint8))
    assert a.compare_to_fingerprint(b, 200) == 0.0


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU behaviour")
def test_compute_calls_fail_loudly_without_gpu(lb):
    """No CPU fallback: every compute entry point reports kLBAudioDetectiveDeviceUnavailable."""
    nogp = lb.constant("kLBAudioDetectiveDeviceUnavailable")
    d = lb.Detective().configure(sample_rate=44100, window=1024)
    with pytest.raises(lb.LBAudioDetectiveError) as e:
        d.process_pcm(np.zeros(44100, np.float32))
    assert e.value.status == nogp
    with pytest.raises(lb.LBAudioDetectiveError):
        d.fingerprint_clips(np.zeros((2, 44100), np.float32))
    a = lb.Fingerprint.from_bools(np.ones((1, 200), np.uint8))
    assert np.isnan(a.compare_to_fingerprint(a, 200))
    with pytest.raises(lb.LBAudioDetectiveError):
        lb.Corpus(200, 5, 10)
    assert lb.lib().LBAudioDetectiveDeviceCount() == 0
    # round 4's additions: layout queries have no answer without a device (the plan lives there), null handles are refused
    assert lb.compact_layout(d) is None and lb.compact_bands(d) is None
    L = lb.lib()
    assert L.LBAudioDetectiveCorpusSetBoundPruning(None, 1) == lb.constant("kLBAudioDetectiveArgumentInvalid")
    assert L.LBAudioDetectiveSetFilePipeline(None, 1) != 0
    n = C.c_uint32(7)
    assert L.LBAudioDetectiveGetCompactBands(None, None, C.byref(n)) != 0


def test_missing_library_raises(lb, monkeypatch, tmp_path):
    from lbaudiodetective_amd import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "absent.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _native.lib()


def test_unsupported_and_missing_files(lb, tmp_path):
    d = lb.Detective()
    out = lb._native.Ref()
    assert lb.lib().LBAudioDetectiveProcessAudioURL(d._ref, None, C.byref(out)) == 1     # nil URL (D.m:211-214)
    assert lb.lib().LBAudioDetectiveProcessAudioURL(d._ref, str(tmp_path / "nope.caf").encode(), C.byref(out)) == -43
    p = tmp_path / "junk.caf"
    p.write_bytes(b"not audio at all")
    assert lb.lib().LBAudioDetectiveProcessAudioURL(d._ref, str(p).encode(), C.byref(out)) == \
        lb.constant("kLBAudioDetectiveUnsupportedFile")


def _build_example(tmp_path, lb, name="compare_urls"):
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / name)
    libdir = os.path.dirname(lb.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-D_DEFAULT_SOURCE", "-Wall", "-Wextra", "-Werror", "-I",
                           os.path.join(root, "include"), os.path.join(root, "examples", name + ".c"), "-L", libdir,
                           "-llbaudiodetective", "-Wl,-rpath," + libdir, "-o", exe])
    return exe


def test_c_host_builds_against_the_header(lb, tmp_path):
    """A C99 program written like the upstream README snippet compiles against include/ and links the
    library: the boundary really is plain C."""
    import subprocess
    exe = _build_example(tmp_path, lb)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 2 and "lbaudiodetective-amd" in out.stderr
    if not _has_gpu():
        birds = os.path.join(os.path.dirname(__file__), "golden", "birds")
        out = subprocess.run([exe, os.path.join(birds, "BlackBird.caf"), os.path.join(birds, "BlackBird_eql.caf")],
                             capture_output=True, text=True)
        assert out.returncode == 1 and "OSStatus" in out.stderr       # 'nogp': no fallback


def test_sharded_c_host_builds_and_fails_loudly_without_gpu(lb, tmp_path):
    """examples/sharded_query.c (one process per GPU, RCCL behind the C ABI, no Python) is plain C99 too; without a
    GPU its first device call reports an OSStatus instead of computing anything."""
    import subprocess
    exe = _build_example(tmp_path, lb, "sharded_query")
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 2 and "usage" in out.stderr
    if not _has_gpu():
        out = subprocess.run([exe, "0", "1", str(tmp_path / "id"), "1000"], capture_output=True, text=True)
        assert out.returncode == 1 and "OSStatus" in out.stderr


def test_header_compiles_for_an_objective_c_host():
    """north_star: "Objective-C host -> HIP".  tests/objc_host_snippet.m passes NSURL* to LBAudioDetectiveProcessAudioURL /
    ...CompareAudioURLs exactly like LBAudioDetectiveTests.m:66-68; the header must compile as Objective-C (manual
    retain/release and ARC) and Objective-C++ without any Apple SDK header."""
    import shutil, subprocess
    clang = shutil.which("clang") or "/opt/rocm/lib/llvm/bin/clang"
    if not os.path.exists(clang):
        pytest.skip("no clang")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "tests", "objc_host_snippet.m")
    for lang, extra in (("objective-c", []), ("objective-c", ["-fobjc-arc", "-fobjc-runtime=macosx-10.13"]), ("objective-c++", [])):
        out = subprocess.run([clang, "-x", lang, *extra, "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I",
                              os.path.join(root, "include"), src], capture_output=True, text=True)
        assert out.returncode == 0, (lang, extra, out.stderr[-1500:])


def test_out_of_range_indices_are_safe(lb):
    fp = lb.Fingerprint.from_bools(np.ones((2, 8), np.uint8))
    assert not fp.subfingerprint_at_index(5).any()
    fr = lb.Frame(2)
    assert not fr.set_row([1, 2], 7) and fr.number_of_rows == 0
