"""Python mirror of the LBAudioDetective C interface over liblbaudiodetective.so.

Class and method names follow the upstream functions (``LBAudioDetectiveFingerprintCompareToFingerprint``
-> ``Fingerprint.compare_to_fingerprint``), argument meaning and error behaviour are the C
library's.  Everything that computes goes through the C ABI into HIP kernels; this module only
marshals buffers (numpy on the host, ``torch`` tensors for device memory and streams).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N

noErr = 0


class LBAudioDetectiveError(RuntimeError):
    def __init__(self, status: int, what: str):
        self.status = status
        code = status & 0xFFFFFFFF
        four = bytes([(code >> s) & 0xFF for s in (24, 16, 8, 0)])
        tag = f"'{four.decode()}'" if all(32 <= b < 127 for b in four) else str(status)
        super().__init__(f"{what}: OSStatus {tag}")


def _check(status: int, what: str):
    if status != noErr:
        raise LBAudioDetectiveError(status, what)


def _u8(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint8)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _stream_ptr(stream=None):
    """hipStream_t of a torch stream (default: torch's current stream; None without torch)."""
    if stream is not None:
        return C.c_void_p(stream.cuda_stream)
    try:
        import torch
        if torch.cuda.is_available():
            return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    except ImportError:
        pass
    return C.c_void_p(0)


def pack_subfingerprint(bools) -> np.ndarray:
    b = _u8(bools)
    out = np.zeros(N.PACKED_WORDS, np.uint32)
    N.lib().LBAudioDetectivePackSubfingerprint(b.ctypes.data, b.size, out.ctypes.data)
    return out


def unpack_subfingerprint(words, length: int) -> np.ndarray:
    w = np.ascontiguousarray(words, dtype=np.uint32)
    out = np.zeros(length, np.uint8)
    N.lib().LBAudioDetectiveUnpackSubfingerprint(w.ctypes.data, length, out.ctypes.data)
    return out


def unpack_packed(packed: np.ndarray, length: int) -> np.ndarray:
    """[..., 8] uint32 (or [..., 32] uint8) packed sub-fingerprints -> [..., length] Booleans."""
    p = np.ascontiguousarray(packed)
    if p.dtype == np.uint8:
        p = p.view(np.uint32)
    p = p.reshape(-1, N.PACKED_WORDS)
    bits = ((p[:, :, None] >> np.arange(32, dtype=np.uint32)[None, None, :]) & 1).astype(np.uint8)
    return bits.reshape(p.shape[0], 256)[:, :length]


class Fingerprint:
    """LBAudioDetectiveFingerprintRef."""

    def __init__(self, subfingerprint_length: int = 0, _ref=None):
        self._L = N.lib()
        self._ref = _ref if _ref is not None else self._L.LBAudioDetectiveFingerprintNew(subfingerprint_length)

    @classmethod
    def from_bools(cls, bools) -> "Fingerprint":
        b = _u8(bools)
        assert b.ndim == 2
        fp = cls(b.shape[1])
        for row in b:
            fp.add_subfingerprint(row)
        return fp

    def dispose(self):
        if self._ref:
            self._L.LBAudioDetectiveFingerprintDispose(self._ref)
            self._ref = None

    def __del__(self):
        try:
            self.dispose()
        except Exception:
            pass

    def copy(self) -> "Fingerprint":
        return Fingerprint(_ref=self._L.LBAudioDetectiveFingerprintCopy(self._ref))

    @property
    def subfingerprint_length(self) -> int:
        return self._L.LBAudioDetectiveFingerprintGetSubfingerprintLength(self._ref)

    @property
    def number_of_subfingerprints(self) -> int:
        return self._L.LBAudioDetectiveFingerprintGetNumberOfSubfingerprints(self._ref)

    def subfingerprint_at_index(self, index: int) -> np.ndarray:
        out = np.zeros(self.subfingerprint_length, np.uint8)
        self._L.LBAudioDetectiveFingerprintGetSubfingerprintAtIndex(self._ref, index, out.ctypes.data)
        return out

    def set_subfingerprint_length(self, length: int):
        io = N.UInt32(length)
        ok = self._L.LBAudioDetectiveFingerprintSetSubfingerprintLength(self._ref, C.byref(io))
        return bool(ok), io.value

    def add_subfingerprint(self, bools):
        b = _u8(bools)
        assert b.size >= self.subfingerprint_length
        self._L.LBAudioDetectiveFingerprintAddSubfingerprint(self._ref, b.ctypes.data)

    def equal_to_fingerprint(self, other: "Fingerprint") -> bool:
        return bool(self._L.LBAudioDetectiveFingerprintEqualToFingerprint(self._ref, other._ref))

    def compare_to_fingerprint(self, other: "Fingerprint", range_: int) -> float:
        return float(self._L.LBAudioDetectiveFingerprintCompareToFingerprint(self._ref, other._ref, range_))

    def compare_subfingerprints(self, a, b, range_: int) -> float:
        a, b = _u8(a), _u8(b)
        return float(self._L.LBAudioDetectiveFingerprintCompareSubfingerprints(self._ref, a.ctypes.data, b.ctypes.data,
                                                                                range_))

    def to_bools(self) -> np.ndarray:
        n, L = self.number_of_subfingerprints, self.subfingerprint_length
        out = np.zeros((n, L), np.uint8)
        for i in range(n):
            out[i] = self.subfingerprint_at_index(i)
        return out

    def to_string(self) -> str:
        """'0'/'1' per Boolean, sub-fingerprints joined by '+' (LBAudioDetectiveTests.m:22-37)."""
        n = int(self._L.LBAudioDetectiveFingerprintGetStringLength(self._ref))
        buf = C.create_string_buffer(n + 1)
        self._L.LBAudioDetectiveFingerprintGetString(self._ref, buf, n + 1)
        return buf.value.decode()

    @classmethod
    def from_string(cls, text: str) -> "Fingerprint":
        ref = N.lib().LBAudioDetectiveFingerprintNewFromString(text.encode())
        if not ref:
            raise ValueError("malformed fingerprint string")
        return cls(_ref=ref)


class Frame:
    """LBAudioDetectiveFrameRef."""

    def __init__(self, max_row_count: int, _ref=None):
        self._L = N.lib()
        self._ref = _ref if _ref is not None else self._L.LBAudioDetectiveFrameNew(max_row_count)

    def dispose(self):
        if self._ref:
            self._L.LBAudioDetectiveFrameDispose(self._ref)
            self._ref = None

    def __del__(self):
        try:
            self.dispose()
        except Exception:
            pass

    def copy(self) -> "Frame":
        return Frame(0, _ref=self._L.LBAudioDetectiveFrameCopy(self._ref))

    @property
    def number_of_rows(self) -> int:
        return self._L.LBAudioDetectiveFrameGetNumberOfRows(self._ref)

    def full(self) -> bool:
        return bool(self._L.LBAudioDetectiveFrameFull(self._ref))

    def set_row(self, row, index: int) -> bool:
        r = _f32(row)
        return bool(self._L.LBAudioDetectiveFrameSetRow(self._ref, r.ctypes.data, index, r.size))

    def get_value(self, row: int, col: int) -> float:
        return float(self._L.LBAudioDetectiveFrameGetValue(self._ref, row, col))

    def get_row(self, index: int, count: int) -> np.ndarray:
        p = self._L.LBAudioDetectiveFrameGetRow(self._ref, index)
        return np.ctypeslib.as_array(p, shape=(count,)).copy()

    def decompose(self):
        self._L.LBAudioDetectiveFrameDecompose(self._ref)

    def fingerprint_length(self) -> int:
        return self._L.LBAudioDetectiveFrameFingerprintLength(self._ref)

    def fingerprint_size(self) -> int:
        return self._L.LBAudioDetectiveFrameFingerprintSize(self._ref)

    def extract_fingerprint(self, n_wavelets: int) -> np.ndarray:
        out = np.zeros(2 * n_wavelets, np.uint8)
        self._L.LBAudioDetectiveFrameExtractFingerprint(self._ref, n_wavelets, out.ctypes.data)
        return out

    def equal_to_frame(self, other: "Frame") -> bool:
        return bool(self._L.LBAudioDetectiveFrameEqualToFrame(self._ref, other._ref))


class Detective:
    """LBAudioDetectiveRef."""

    def __init__(self):
        self._L = N.lib()
        self._ref = self._L.LBAudioDetectiveNew()

    def dispose(self):
        if self._ref:
            self._L.LBAudioDetectiveDispose(self._ref)
            self._ref = None

    def __del__(self):
        try:
            self.dispose()
        except Exception:
            pass

    # getters / setters (D.h:74-205)
    processing_sample_rate = property(
        lambda s: s._L.LBAudioDetectiveGetProcessingSampleRate(s._ref),
        lambda s, v: _check(s._L.LBAudioDetectiveSetProcessingSampleRate(s._ref, float(v)), "SetProcessingSampleRate"))
    number_of_pitch_steps = property(
        lambda s: s._L.LBAudioDetectiveGetNumberOfPitchSteps(s._ref),
        lambda s, v: _check(s._L.LBAudioDetectiveSetNumberOfPitchSteps(s._ref, int(v)), "SetNumberOfPitchSteps"))
    subfingerprint_length = property(
        lambda s: s._L.LBAudioDetectiveGetSubfingerprintLength(s._ref),
        lambda s, v: _check(s._L.LBAudioDetectiveSetSubfingerprintLength(s._ref, int(v)), "SetSubfingerprintLength"))
    window_size = property(
        lambda s: s._L.LBAudioDetectiveGetWindowSize(s._ref),
        lambda s, v: _check(s._L.LBAudioDetectiveSetWindowSize(s._ref, int(v)), "SetWindowSize"))
    analysis_stride = property(
        lambda s: s._L.LBAudioDetectiveGetAnalysisStride(s._ref),
        lambda s, v: _check(s._L.LBAudioDetectiveSetAnalysisStride(s._ref, int(v)), "SetAnalysisStride"))

    def set_window_size_status(self, v: int) -> int:
        return self._L.LBAudioDetectiveSetWindowSize(self._ref, int(v))

    def configure(self, sample_rate=None, window=None, stride=None, bands=None, subfp_len=None) -> "Detective":
        if sample_rate is not None:
            self.processing_sample_rate = sample_rate
        if window is not None:
            self.window_size = window
        if stride is not None:
            self.analysis_stride = stride
        if bands is not None:
            self.number_of_pitch_steps = bands
        if subfp_len is not None:
            self.subfingerprint_length = subfp_len
        return self

    def set_file_pipeline(self, enabled: bool):
        """Two runs of a file batch in flight (default) or one at a time (LBAudioDetectiveSetFilePipeline)."""
        _check(self._L.LBAudioDetectiveSetFilePipeline(self._ref, 1 if enabled else 0), "SetFilePipeline")
        return self

    def set_kernel_variant(self, variant: int):
        _check(self._L.LBAudioDetectiveSetKernelVariant(self._ref, variant), "SetKernelVariant")

    def set_scratch_limit(self, n_bytes: int):
        _check(self._L.LBAudioDetectiveSetScratchLimit(self._ref, n_bytes), "SetScratchLimit")

    def set_stage_timing(self, enabled: bool):
        _check(self._L.LBAudioDetectiveSetStageTiming(self._ref, int(enabled)), "SetStageTiming")

    def stage_times(self):
        """(stage 1 ms, stage 2 ms, launches per stage) of the last batch call; waits for it."""
        a, b, n = N.Float32(0), N.Float32(0), N.UInt32(0)
        _check(self._L.LBAudioDetectiveGetStageTimes(self._ref, C.byref(a), C.byref(b), C.byref(n)), "GetStageTimes")
        return float(a.value), float(b.value), int(n.value)

    def subfingerprint_count(self, n_samples: int) -> int:
        return int(self._L.LBAudioDetectiveGetSubfingerprintCount(self._ref, n_samples))

    def set_kernel_tuning(self, waves_per_workgroup: int = 0, twiddle_cache: bool = True):
        """Measurement knobs of the generic stage-1 kernel (tools/sweep_lds_tiles.py)."""
        _check(self._L.LBAudioDetectiveSetKernelTuning(self._ref, waves_per_workgroup, int(twiddle_cache)), "SetKernelTuning")
        return self

    def set_file_hop_mode(self, mode: int):
        """1 (default): upstream's file-frame hop; 0: hop in processing-rate samples."""
        _check(self._L.LBAudioDetectiveSetFileHopMode(self._ref, mode), "SetFileHopMode")
        return self

    def set_file_tail_mode(self, mode: int):
        """Hop mode 1, windows reaching past the end of the file: 1 (default) nothing is read -> zero rows,
        2 partial reads over the stale spectrum, 0 zero-filled."""
        _check(self._L.LBAudioDetectiveSetFileTailMode(self._ref, mode), "SetFileTailMode")
        return self

    def set_resampler_mode(self, mode: int):
        """0 (default) long Kaiser sinc, 1 short sinc, 2 linear interpolation."""
        _check(self._L.LBAudioDetectiveSetResamplerMode(self._ref, mode), "SetResamplerMode")
        return self

    def process_file_stream(self, client_samples, file_frames: int, hop: int) -> "Fingerprint":
        """Upstream's file loop (D.m:241-293) on a file already converted to the processing rate."""
        x = _f32(client_samples).reshape(-1)
        out = N.Ref()
        _check(self._L.LBAudioDetectiveProcessFileStream(self._ref, x.ctypes.data, x.size, int(file_frames), int(hop),
                                                         C.byref(out)), "ProcessFileStream")
        return Fingerprint(_ref=out.value)

    # file entry points (D.h:218,235)
    def process_audio_url(self, path: str) -> Fingerprint:
        out = N.Ref()
        _check(self._L.LBAudioDetectiveProcessAudioURL(self._ref, path.encode(), C.byref(out)), "ProcessAudioURL")
        return Fingerprint(_ref=out.value)

    def process_audio_urls(self, paths, statuses: bool = False):
        """LBAudioDetectiveProcessAudioURLs: every file through one launch chain -> list of Fingerprint (None where a
        file failed; with statuses=True also the list of OSStatus values, else a failure raises)."""
        n = len(paths)
        arr = (C.c_char_p * n)(*[p.encode() for p in paths])
        refs = (N.Ref * n)()
        sts = (N.OSStatus * n)()
        _check(self._L.LBAudioDetectiveProcessAudioURLs(self._ref, arr, n, refs, sts), "ProcessAudioURLs")
        fps = [Fingerprint(_ref=refs[i]) if refs[i] else None for i in range(n)]
        if statuses:
            return fps, [int(sts[i]) for i in range(n)]
        for i in range(n):
            _check(int(sts[i]), f"ProcessAudioURLs[{paths[i]}]")
        return fps

    def convert_audio_url(self, path: str):
        """The device front end alone: (mono samples at the processing rate, file frames, file rate)."""
        buf, n, ff, rate = C.POINTER(N.Float32)(), N.UInt64(0), N.UInt64(0), N.Float64(0.0)
        _check(self._L.LBAudioDetectiveConvertAudioURL(self._ref, path.encode(), C.byref(buf), C.byref(n), C.byref(ff),
                                                       C.byref(rate)), "ConvertAudioURL")
        try:
            out = np.ctypeslib.as_array(buf, shape=(n.value,)).copy() if n.value else np.zeros(0, np.float32)
        finally:
            self._L.LBAudioDetectiveFreeSamples(buf)
        return out, int(ff.value), float(rate.value)

    def compare_audio_urls(self, path1: str, path2: str, range_: int = 0) -> float:
        m = N.Float32(float("nan"))
        _check(self._L.LBAudioDetectiveCompareAudioURLs(self._ref, path1.encode(), path2.encode(), range_, C.byref(m)),
               "CompareAudioURLs")
        return float(m.value)

    # PCM entry points
    def process_pcm(self, pcm) -> Fingerprint:
        x = _f32(pcm).reshape(-1)
        out = N.Ref()
        _check(self._L.LBAudioDetectiveProcessPCM(self._ref, x.ctypes.data, x.size, C.byref(out)), "ProcessPCM")
        return Fingerprint(_ref=out.value)

    def compare_pcm(self, pcm1, pcm2, range_: int = 0) -> float:
        a, b = _f32(pcm1).reshape(-1), _f32(pcm2).reshape(-1)
        m = N.Float32(float("nan"))
        _check(self._L.LBAudioDetectiveComparePCM(self._ref, a.ctypes.data, a.size, b.ctypes.data, b.size, range_,
                                                  C.byref(m)), "ComparePCM")
        return float(m.value)

    def fingerprint_clips(self, clips) -> np.ndarray:
        """Host batch: [n_clips, samples] float32 (or int16 / int32 PCM) -> [n_clips, count, subfp_len] Booleans."""
        x = np.asarray(clips)
        fmt = {np.dtype(np.int16): 1, np.dtype(np.int32): 2}.get(x.dtype, 0)
        x = np.ascontiguousarray(x) if fmt else _f32(x)
        n, spc = x.shape
        per = self.subfingerprint_count(spc)
        out = np.zeros((n, per, self.subfingerprint_length), np.uint8)
        _check(self._L.LBAudioDetectiveFingerprintClipsFormat(self._ref, x.ctypes.data, fmt, n, spc, out.ctypes.data),
               "FingerprintClips")
        return out

    def fingerprint_clips_device(self, clips, out=None, stream=None, taps: bool = False):
        """Device batch on torch tensors: clips [n, samples] float32 (cuda) -> packed uint8 [n, count, 32].

        Asynchronous on the given (default: current) torch stream.  With taps=True also returns the
        128 x bands frames before and after the Haar (unfused kernels only)."""
        import torch
        assert clips.is_cuda and clips.is_contiguous() and clips.dtype in (torch.float32, torch.int16, torch.int32)
        assert not taps or clips.dtype == torch.float32
        n, spc = clips.shape
        per = self.subfingerprint_count(spc)
        if out is None:
            out = torch.empty((n, per, N.PACKED_BYTES), dtype=torch.uint8, device=clips.device)
        sp = _stream_ptr(stream)
        if not taps:
            fmt = {torch.float32: 0, torch.int16: 1, torch.int32: 2}[clips.dtype]
            _check(self._L.LBAudioDetectiveFingerprintClipsDeviceFormat(self._ref, clips.data_ptr(), fmt, n, spc,
                                                                       out.data_ptr(), sp), "FingerprintClipsDevice")
            return out
        bands = self.number_of_pitch_steps
        raw = torch.empty((n, per, N.ROWS_PER_FRAME, bands), dtype=torch.float32, device=clips.device)
        haar = torch.empty_like(raw)
        _check(self._L.LBAudioDetectiveFingerprintClipsDeviceTaps(self._ref, clips.data_ptr(), n, spc, out.data_ptr(),
                                                                 raw.data_ptr(), haar.data_ptr(), sp),
               "FingerprintClipsDeviceTaps")
        return out, raw, haar


def compact_layout(det: "Detective"):
    """(live band among bands 0..15 or 32, columns of the row transform that can be non-zero) of the configuration's compact
    inter-stage frames, or None when it has none (LBAudioDetectiveGetCompactLayout)."""
    left, cols = N.UInt32(0), N.UInt32(0)
    st = N.lib().LBAudioDetectiveGetCompactLayout(det._ref, C.byref(left), C.byref(cols))
    return (int(left.value), int(cols.value)) if st == 0 else None


def compact_bands(det: "Detective"):
    """The bands a row of a compact inter-stage frame holds, in storage order (LBAudioDetectiveGetCompactBands), or None."""
    bands, count = (N.UInt32 * 17)(), N.UInt32(0)
    st = N.lib().LBAudioDetectiveGetCompactBands(det._ref, bands, C.byref(count))
    return [int(bands[i]) for i in range(count.value)] if st == 0 else None


def frames_to_subfingerprints_device(det: "Detective", frames, want_haar: bool = False, stream=None, compact: bool = False):
    """Stage 2 alone on torch frames [n, 128, bands] float32 (cuda) -> packed uint8 [n, 32] (and the Haar frames).
    compact: through the SPARSE form -- the frames (whose structurally empty bands must be zero) are packed into the
    compact layout first (128 rows of the bands that can be non-zero)."""
    import torch
    assert frames.is_cuda and frames.dtype == torch.float32 and frames.is_contiguous()
    n = frames.shape[0]
    out = torch.empty((n, N.PACKED_BYTES), dtype=torch.uint8, device=frames.device)
    haar = torch.empty_like(frames) if want_haar else None
    if compact:
        lay = compact_layout(det)
        if lay is None:
            raise LBAudioDetectiveError(1, "this configuration has no compact frame layout")
        cf = frames[:, :, compact_bands(det)].contiguous()          # [n, 128, stored bands]
        _check(N.lib().LBAudioDetectiveCompactFramesToSubfingerprintsDevice(det._ref, cf.data_ptr(), n, out.data_ptr(),
                                                                            haar.data_ptr() if want_haar else None, _stream_ptr(stream)),
               "CompactFramesToSubfingerprintsDevice")
        return (out, haar) if want_haar else out
    _check(N.lib().LBAudioDetectiveFramesToSubfingerprintsDevice(det._ref, frames.data_ptr(), n, out.data_ptr(),
                                                                 haar.data_ptr() if want_haar else None, _stream_ptr(stream)),
           "FramesToSubfingerprintsDevice")
    return (out, haar) if want_haar else out


class Stream:
    """LBAudioDetectiveStreamRef: chunked PCM in, partial frame carried across calls."""

    def __init__(self, detective: Detective):
        self._L = N.lib()
        self._det = detective            # keep the detective alive
        self._ref = self._L.LBAudioDetectiveStreamNew(detective._ref)

    def dispose(self):
        if self._ref:
            self._L.LBAudioDetectiveStreamDispose(self._ref)
            self._ref = None

    def __del__(self):
        try:
            self.dispose()
        except Exception:
            pass

    def push(self, samples) -> int:
        x = _f32(samples).reshape(-1)
        new = N.UInt32(0)
        _check(self._L.LBAudioDetectiveStreamPush(self._ref, x.ctypes.data, x.size, C.byref(new)), "StreamPush")
        return int(new.value)

    def fingerprint(self) -> Fingerprint:
        return Fingerprint(_ref=self._L.LBAudioDetectiveStreamCopyFingerprint(self._ref))


class Corpus:
    """LBAudioDetectiveCorpusRef: device-resident reference fingerprints with a top-1 query."""

    def __init__(self, subfingerprint_length: int, subfingerprints_per_entry: int, capacity: int, _ref=None):
        self._L = N.lib()
        self._ref = _ref if _ref is not None else self._L.LBAudioDetectiveCorpusNew(
            subfingerprint_length, subfingerprints_per_entry, capacity)
        if not self._ref:
            raise LBAudioDetectiveError(1, "CorpusNew (unsupported shape, zero capacity or no HIP device)")
        self.subfingerprint_length = subfingerprint_length
        self.subfingerprints_per_entry = subfingerprints_per_entry
        self.capacity = capacity

    @classmethod
    def ragged(cls, subfingerprint_length: int, entry_capacity: int, subfingerprint_capacity: int) -> "Corpus":
        """LBAudioDetectiveCorpusNewRagged: entries of any length (the shape of LBAudioDetectiveTests.m:57-91)."""
        ref = N.lib().LBAudioDetectiveCorpusNewRagged(subfingerprint_length, entry_capacity, subfingerprint_capacity)
        if not ref:
            raise LBAudioDetectiveError(1, "CorpusNewRagged (unsupported length, zero capacity or no HIP device)")
        return cls(subfingerprint_length, 0, entry_capacity, _ref=ref)

    def set_bound_pruning(self, enabled: bool):
        """Ragged corpora, top-1 queries: drop groups of sliding offsets that cannot reach the best match found so far
        (exact; on by default).  LBAudioDetectiveCorpusSetBoundPruning."""
        _check(self._L.LBAudioDetectiveCorpusSetBoundPruning(self._ref, 1 if enabled else 0), "CorpusSetBoundPruning")
        return self

    def set_bound_pruning_threshold(self, score: float):
        """The score from which a match is published and bounds the rest of a top-1 scan (default 0.7)."""
        _check(self._L.LBAudioDetectiveCorpusSetBoundPruningThreshold(self._ref, float(score)), "CorpusSetBoundPruningThreshold")
        return self

    @property
    def bound_pruning_threshold(self) -> float:
        return float(self._L.LBAudioDetectiveCorpusGetBoundPruningThreshold(self._ref))

    @property
    def subfingerprint_total(self) -> int:
        return int(self._L.LBAudioDetectiveCorpusGetSubfingerprintTotal(self._ref))

    def append_ragged_packed_device(self, packed, counts, stream=None):
        """packed: torch uint8 [sum(counts), 32] on the device; counts: the entries' sub-fingerprint counts (host)."""
        assert packed.is_cuda and packed.is_contiguous()
        cnt = np.ascontiguousarray(counts, dtype=np.uint32)
        assert int(cnt.sum()) == packed.shape[0]
        _check(self._L.LBAudioDetectiveCorpusAppendRaggedPackedDevice(self._ref, packed.data_ptr(), cnt.ctypes.data,
                                                                     cnt.size, _stream_ptr(stream)),
               "CorpusAppendRaggedPackedDevice")

    def dispose(self):
        if self._ref:
            self._L.LBAudioDetectiveCorpusDispose(self._ref)
            self._ref = None

    def __del__(self):
        try:
            self.dispose()
        except Exception:
            pass

    def __len__(self) -> int:
        return int(self._L.LBAudioDetectiveCorpusGetCount(self._ref))

    def save(self, path: str):
        _check(self._L.LBAudioDetectiveCorpusSave(self._ref, path.encode()), "CorpusSave")

    @classmethod
    def load(cls, path: str, subfingerprint_length: int, subfingerprints_per_entry: int, capacity: int = 0) -> "Corpus":
        ref = N.lib().LBAudioDetectiveCorpusLoad(path.encode(), capacity)
        if not ref:
            raise LBAudioDetectiveError(1, "CorpusLoad (missing/malformed file or no HIP device)")
        c = cls(subfingerprint_length, subfingerprints_per_entry, capacity, _ref=ref)
        return c

    @property
    def entry_stride_bytes(self) -> int:
        return int(self._L.LBAudioDetectiveCorpusGetEntryStrideBytes(self._ref))

    def set_kernel_variant(self, variant: int):
        _check(self._L.LBAudioDetectiveCorpusSetKernelVariant(self._ref, variant), "CorpusSetKernelVariant")

    def append_packed_device(self, packed, stream=None):
        """packed: torch uint8 [n, per_entry, 32] on the device."""
        assert packed.is_cuda and packed.is_contiguous()
        n = packed.shape[0]
        _check(self._L.LBAudioDetectiveCorpusAppendPackedDevice(self._ref, packed.data_ptr(), n, _stream_ptr(stream)),
               "CorpusAppendPackedDevice")

    def append_fingerprint(self, fp: Fingerprint):
        _check(self._L.LBAudioDetectiveCorpusAppendFingerprint(self._ref, fp._ref), "CorpusAppendFingerprint")

    def query(self, fp: Fingerprint, range_: int = 0):
        idx, score = N.SInt64(-1), N.Float32(0.0)
        _check(self._L.LBAudioDetectiveCorpusQuery(self._ref, fp._ref, range_, C.byref(idx), C.byref(score)), "CorpusQuery")
        return int(idx.value), float(score.value)

    def query_batch(self, fps, range_: int = 0):
        """Several queries in one pass over the corpus -> list of (index, score)."""
        n = len(fps)
        refs = (N.Ref * n)(*[f._ref for f in fps])
        idx, sc = (N.SInt64 * n)(), (N.Float32 * n)()
        _check(self._L.LBAudioDetectiveCorpusQueryBatch(self._ref, refs, n, range_, idx, sc), "CorpusQueryBatch")
        return [(int(idx[i]), float(sc[i])) for i in range(n)]

    def query_batch_keys_device(self, fps, keys_out, range_: int = 0, index_base: int = 0, stream=None):
        """Writes len(fps) 64-bit keys into keys_out (torch int64 on the device), asynchronously."""
        n = len(fps)
        refs = (N.Ref * n)(*[f._ref for f in fps])
        _check(self._L.LBAudioDetectiveCorpusQueryBatchKeysDevice(self._ref, refs, n, range_, index_base,
                                                                 keys_out.data_ptr(), _stream_ptr(stream)),
               "CorpusQueryBatchKeysDevice")
        return keys_out

    def query_key_device(self, fp: Fingerprint, key_out, range_: int = 0, index_base: int = 0, stream=None):
        """Writes the 64-bit (score, ~index) key into key_out (torch int64[1] on the device), async."""
        _check(self._L.LBAudioDetectiveCorpusQueryKeyDevice(self._ref, fp._ref, range_, index_base, key_out.data_ptr(),
                                                           _stream_ptr(stream)), "CorpusQueryKeyDevice")
        return key_out

    def query_sharded(self, fp: Fingerprint, comm: "Comm", index_base: int = 0, range_: int = 0, stream=None):
        """LBAudioDetectiveCorpusQuerySharded: this rank's scan + the library's own RCCL all-reduce (ncclUint64,
        ncclMax) of the key; collective, every rank gets (global index, score)."""
        idx, score = N.SInt64(-1), N.Float32(0.0)
        _check(self._L.LBAudioDetectiveCorpusQuerySharded(self._ref, fp._ref, range_, index_base, comm._ref,
                                                         _stream_ptr(stream), C.byref(idx), C.byref(score)),
               "CorpusQuerySharded")
        return int(idx.value), float(score.value)

    def query_batch_sharded(self, fps, comm: "Comm", index_base: int = 0, range_: int = 0, stream=None):
        n = len(fps)
        refs = (N.Ref * n)(*[f._ref for f in fps])
        idx, sc = (N.SInt64 * n)(), (N.Float32 * n)()
        _check(self._L.LBAudioDetectiveCorpusQueryBatchSharded(self._ref, refs, n, range_, index_base, comm._ref,
                                                              _stream_ptr(stream), idx, sc), "CorpusQueryBatchSharded")
        return [(int(idx[i]), float(sc[i])) for i in range(n)]

    def query_batch_sharded_with(self, fps, all_reduce, index_base: int = 0, range_: int = 0, stream=None, context=None):
        """LBAudioDetectiveCorpusQueryBatchShardedWith: the sharded query with the caller's exchange step --
        all_reduce(context, device_keys_ptr, count, stream_ptr) -> status must leave the element-wise MAX over all
        ranks in the `count` unsigned 64-bit keys at device_keys_ptr.  `self` may be None-like (pass corpus=None through
        sharded_query_without_corpus) only from tests of the failing-rank path."""
        n = len(fps)
        refs = (N.Ref * n)(*[f._ref for f in fps])
        idx, sc = (N.SInt64 * n)(), (N.Float32 * n)()
        cb = all_reduce if isinstance(all_reduce, N.AllReduceMaxFn) else N.AllReduceMaxFn(all_reduce)
        st = self._L.LBAudioDetectiveCorpusQueryBatchShardedWith(self._ref, refs, n, range_, index_base, cb, context,
                                                                _stream_ptr(stream), idx, sc)
        _check(st, "CorpusQueryBatchShardedWith")
        return [(int(idx[i]), float(sc[i])) for i in range(n)]

    def scores_device(self, fp: Fingerprint, range_: int = 0, stream=None):
        import torch
        out = torch.empty(len(self), dtype=torch.float32, device="cuda")
        _check(self._L.LBAudioDetectiveCorpusScoresDevice(self._ref, fp._ref, range_, out.data_ptr(), _stream_ptr(stream)),
               "CorpusScoresDevice")
        return out

    @staticmethod
    def decode_key(key: int):
        idx, score = N.SInt64(-1), N.Float32(0.0)
        N.lib().LBAudioDetectiveCorpusDecodeKey(key & 0xFFFFFFFFFFFFFFFF, C.byref(idx), C.byref(score))
        return int(idx.value), float(score.value)


class Comm:
    """An RCCL communicator made through the library's helpers (ncclCommInitRank with the current device).
    `unique_id()` on rank 0, the 128 bytes travel to the other ranks by whatever means the host has, then every
    rank constructs Comm(n_ranks, unique_id, rank)."""

    UNIQUE_ID_BYTES = 128

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(Comm.UNIQUE_ID_BYTES)
        _check(N.lib().LBAudioDetectiveCommGetUniqueId(buf), "CommGetUniqueId")
        return buf.raw

    def __init__(self, n_ranks: int, unique_id: bytes, rank: int):
        assert len(unique_id) == Comm.UNIQUE_ID_BYTES
        self._L = N.lib()
        self.n_ranks, self.rank = n_ranks, rank
        ref = C.c_void_p()
        _check(self._L.LBAudioDetectiveCommInitRank(C.byref(ref), n_ranks, unique_id, rank), "CommInitRank")
        self._ref = ref

    def info(self):
        """(ranks that joined the communicator, this rank's number in it) -- ncclCommCount / ncclCommUserRank."""
        n, r = N.SInt32(0), N.SInt32(0)
        _check(self._L.LBAudioDetectiveCommGetInfo(self._ref, C.byref(n), C.byref(r)), "CommGetInfo")
        return int(n.value), int(r.value)

    def dispose(self):
        if getattr(self, "_ref", None):
            self._L.LBAudioDetectiveCommDestroy(self._ref)
            self._ref = None

    def __del__(self):
        try:
            self.dispose()
        except Exception:
            pass


def read_audio_url(path: str, sample_rate: float = 0.0, resampler: int = 0):
    """Decode a CAF/WAV file to mono float32 (numpy), optionally resampled; returns (samples, rate)."""
    buf, n, rate = C.POINTER(N.Float32)(), N.UInt64(0), N.Float64(0.0)
    _check(N.lib().LBAudioDetectiveReadAudioURLWithResampler(path.encode(), float(sample_rate), int(resampler),
                                                             C.byref(buf), C.byref(n), C.byref(rate)), "ReadAudioURL")
    try:
        out = np.ctypeslib.as_array(buf, shape=(n.value,)).copy() if n.value else np.zeros(0, np.float32)
    finally:
        N.lib().LBAudioDetectiveFreeSamples(buf)
    return out, float(rate.value)


def probe_shader_clock(stream=None, microseconds: int = 2000) -> float:
    """Shader clock in MHz over `microseconds`, measured by a one-wave kernel on `stream` (use a side stream to
    read the clock under another kernel's load); synchronises that stream."""
    mhz = N.Float64(0.0)
    _check(N.lib().LBAudioDetectiveProbeShaderClock(_stream_ptr(stream), microseconds, C.byref(mhz)), "ProbeShaderClock")
    return float(mhz.value)


def synth_clips_device(seed: int, first: int, n_clips: int, sample_rate_hz: int, n_samples: int,
                       stereo_sum: bool = False, out=None, stream=None):
    import torch
    if out is None:
        out = torch.empty((n_clips, n_samples), dtype=torch.float32, device="cuda")
    _check(N.lib().LBAudioDetectiveSynthClipsDevice(seed & 0xFFFFFFFF, first, n_clips, sample_rate_hz, n_samples,
                                                    int(stereo_sum), out.data_ptr(), _stream_ptr(stream)),
           "SynthClipsDevice")
    return out


def synth_ragged_corpus_device(seed: int, first: int, counts, subfp_len: int, stream=None):
    """Synthetic ragged corpus on the device: entry first + e has counts[e] sub-fingerprints (lbo_synth_entry's);
    returns packed uint8 [sum(counts), 32]."""
    import torch
    cnt = np.ascontiguousarray(counts, dtype=np.uint32)
    off = np.zeros(cnt.size + 1, np.uint32)
    np.cumsum(cnt, out=off[1:])
    total = int(off[-1])
    d_off = torch.from_numpy(off.view(np.int32)).cuda()
    out = torch.empty((total, N.PACKED_BYTES), dtype=torch.uint8, device="cuda")
    _check(N.lib().LBAudioDetectiveSynthRaggedCorpusDevice(seed & 0xFFFFFFFF, first, cnt.size, d_off.data_ptr(), total,
                                                           subfp_len, out.data_ptr(), _stream_ptr(stream)),
           "SynthRaggedCorpusDevice")
    torch.cuda.current_stream().synchronize() if stream is None else stream.synchronize()   # d_off is a temporary
    return out


def synth_corpus_device(seed: int, first: int, n_entries: int, n_sub: int, subfp_len: int, out=None, stream=None):
    import torch
    if out is None:
        out = torch.empty((n_entries, n_sub, N.PACKED_BYTES), dtype=torch.uint8, device="cuda")
    _check(N.lib().LBAudioDetectiveSynthCorpusDevice(seed & 0xFFFFFFFF, first, n_entries, n_sub, subfp_len,
                                                     out.data_ptr(), _stream_ptr(stream)), "SynthCorpusDevice")
    return out
