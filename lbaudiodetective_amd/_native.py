"""Loader for liblbaudiodetective.so (the HIP library behind include/lbaudiodetective.h).

There is no fallback of any kind: if the library is missing or fails to load, importing the
product API raises.  `build()` drives hipcc through csrc/Makefile (cross-compiles gfx950 code
objects without a GPU present).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# LBAD_LIB=<path>: another build of the SAME library (the A/B variants of tools/exp/build_variants.sh); never a fallback
LIB_PATH = os.path.abspath(os.environ["LBAD_LIB"]) if os.environ.get("LBAD_LIB") else os.path.join(_HERE, "lib", "liblbaudiodetective.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "lbaudiodetective.h")

# MacTypes
UInt32, SInt32, UInt64, SInt64 = C.c_uint32, C.c_int32, C.c_uint64, C.c_int64
Float32, Float64, Boolean, OSStatus = C.c_float, C.c_double, C.c_ubyte, C.c_int32
Ref = C.c_void_p

PACKED_WORDS = 8
PACKED_BYTES = 32
SHARD_KEYS = 4096          # LBAD_SHARD_KEYS: queries per exchange of a sharded query
ROWS_PER_FRAME = 128


class AudioStreamBasicDescription(C.Structure):
    _fields_ = [
        ("mSampleRate", Float64), ("mFormatID", UInt32), ("mFormatFlags", UInt32),
        ("mBytesPerPacket", UInt32), ("mFramesPerPacket", UInt32), ("mBytesPerFrame", UInt32),
        ("mChannelsPerFrame", UInt32), ("mBitsPerChannel", UInt32), ("mReserved", UInt32),
    ]


def build(force: bool = False, jobs: int = 8) -> str:
    """Compile every HIP source for gfx950 into lib/liblbaudiodetective.so."""
    cmd = ["make", "-C", CSRC, "--no-print-directory", f"-j{jobs}"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"build did not produce {LIB_PATH}")
    return LIB_PATH


_P = C.POINTER
# LBAudioDetectiveAllReduceMaxFn: OSStatus (*)(void* context, UInt64* deviceKeys, UInt32 count, void* stream)
AllReduceMaxFn = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p)

_SIGNATURES = {
    # ---- detective (D.h) ----
    "LBAudioDetectiveNew": (Ref, []),
    "LBAudioDetectiveDispose": (OSStatus, [Ref]),
    "LBAudioDetectiveDefaultProcessingFormat": (AudioStreamBasicDescription, []),
    "LBAudioDetectiveGetProcessingSampleRate": (Float64, [Ref]),
    "LBAudioDetectiveGetNumberOfPitchSteps": (UInt32, [Ref]),
    "LBAudioDetectiveGetSubfingerprintLength": (UInt32, [Ref]),
    "LBAudioDetectiveGetWindowSize": (UInt32, [Ref]),
    "LBAudioDetectiveGetAnalysisStride": (UInt32, [Ref]),
    "LBAudioDetectiveSetProcessingSampleRate": (OSStatus, [Ref, Float64]),
    "LBAudioDetectiveSetNumberOfPitchSteps": (OSStatus, [Ref, UInt32]),
    "LBAudioDetectiveSetSubfingerprintLength": (OSStatus, [Ref, UInt32]),
    "LBAudioDetectiveSetWindowSize": (OSStatus, [Ref, UInt32]),
    "LBAudioDetectiveSetAnalysisStride": (OSStatus, [Ref, UInt32]),
    "LBAudioDetectiveProcessAudioURL": (OSStatus, [Ref, C.c_char_p, _P(Ref)]),
    "LBAudioDetectiveCompareAudioURLs": (OSStatus, [Ref, C.c_char_p, C.c_char_p, UInt32, _P(Float32)]),
    "LBAudioDetectiveProcessAudioPath": (OSStatus, [Ref, C.c_char_p, _P(Ref)]),
    "LBAudioDetectiveCompareAudioPaths": (OSStatus, [Ref, C.c_char_p, C.c_char_p, UInt32, _P(Float32)]),
    "LBAudioDetectiveProcessAudioURLs": (OSStatus, [Ref, _P(C.c_char_p), UInt32, _P(Ref), _P(OSStatus)]),
    "LBAudioDetectiveConvertAudioURL": (OSStatus, [Ref, C.c_char_p, _P(_P(Float32)), _P(UInt64), _P(UInt64), _P(Float64)]),
    # ---- fingerprint (Fp.h) ----
    "LBAudioDetectiveFingerprintNew": (Ref, [UInt32]),
    "LBAudioDetectiveFingerprintDispose": (None, [Ref]),
    "LBAudioDetectiveFingerprintCopy": (Ref, [Ref]),
    "LBAudioDetectiveFingerprintGetSubfingerprintLength": (UInt32, [Ref]),
    "LBAudioDetectiveFingerprintGetNumberOfSubfingerprints": (UInt32, [Ref]),
    "LBAudioDetectiveFingerprintGetSubfingerprintAtIndex": (UInt32, [Ref, UInt32, C.c_void_p]),
    "LBAudioDetectiveFingerprintSetSubfingerprintLength": (Boolean, [Ref, _P(UInt32)]),
    "LBAudioDetectiveFingerprintAddSubfingerprint": (None, [Ref, C.c_void_p]),
    "LBAudioDetectiveFingerprintEqualToFingerprint": (Boolean, [Ref, Ref]),
    "LBAudioDetectiveFingerprintCompareToFingerprint": (Float32, [Ref, Ref, UInt32]),
    "LBAudioDetectiveFingerprintCompareSubfingerprints": (Float32, [Ref, C.c_void_p, C.c_void_p, UInt32]),
    "LBAudioDetectiveFingerprintGetStringLength": (UInt64, [Ref]),
    "LBAudioDetectiveFingerprintGetString": (UInt64, [Ref, C.c_char_p, UInt64]),
    "LBAudioDetectiveFingerprintNewFromString": (Ref, [C.c_char_p]),
    # ---- frame (Fr.h) ----
    "LBAudioDetectiveFrameNew": (Ref, [UInt32]),
    "LBAudioDetectiveFrameDispose": (None, [Ref]),
    "LBAudioDetectiveFrameCopy": (Ref, [Ref]),
    "LBAudioDetectiveFrameGetNumberOfRows": (UInt32, [Ref]),
    "LBAudioDetectiveFrameGetRow": (_P(Float32), [Ref, UInt32]),
    "LBAudioDetectiveFrameGetValue": (Float32, [Ref, UInt32, UInt32]),
    "LBAudioDetectiveFrameFull": (Boolean, [Ref]),
    "LBAudioDetectiveFrameSetRow": (Boolean, [Ref, C.c_void_p, UInt32, UInt32]),
    "LBAudioDetectiveFrameDecompose": (None, [Ref]),
    "LBAudioDetectiveFrameFingerprintSize": (C.c_size_t, [Ref]),
    "LBAudioDetectiveFrameFingerprintLength": (UInt32, [Ref]),
    "LBAudioDetectiveFrameExtractFingerprint": (None, [Ref, UInt32, C.c_void_p]),
    "LBAudioDetectiveFrameEqualToFrame": (Boolean, [Ref, Ref]),
    # ---- additions ----
    "LBAudioDetectiveSetFileHopMode": (OSStatus, [Ref, UInt32]),
    "LBAudioDetectiveSetKernelTuning": (OSStatus, [Ref, UInt32, UInt32]),
    "LBAudioDetectiveSetFileTailMode": (OSStatus, [Ref, UInt32]),
    "LBAudioDetectiveSetResamplerMode": (OSStatus, [Ref, UInt32]),
    "LBAudioDetectiveReadAudioURLWithResampler": (OSStatus, [C.c_char_p, Float64, UInt32, _P(_P(Float32)), _P(UInt64),
                                                            _P(Float64)]),
    "LBAudioDetectiveProcessFileStream": (OSStatus, [Ref, C.c_void_p, UInt64, UInt64, UInt32, _P(Ref)]),
    "LBAudioDetectiveReadAudioURL": (OSStatus, [C.c_char_p, Float64, _P(_P(Float32)), _P(UInt64), _P(Float64)]),
    "LBAudioDetectiveFreeSamples": (None, [_P(Float32)]),
    "LBAudioDetectiveGetSubfingerprintCount": (UInt64, [Ref, UInt64]),
    "LBAudioDetectiveProcessPCM": (OSStatus, [Ref, C.c_void_p, UInt64, _P(Ref)]),
    "LBAudioDetectiveComparePCM": (OSStatus, [Ref, C.c_void_p, UInt64, C.c_void_p, UInt64, UInt32, _P(Float32)]),
    "LBAudioDetectiveFingerprintClipsDevice": (OSStatus, [Ref, C.c_void_p, UInt64, UInt64, C.c_void_p, C.c_void_p]),
    "LBAudioDetectiveFingerprintClipsDeviceFormat": (OSStatus, [Ref, C.c_void_p, UInt32, UInt64, UInt64, C.c_void_p, C.c_void_p]),
    "LBAudioDetectiveSetFilePipeline": (OSStatus, [Ref, UInt32]),
    "LBAudioDetectiveGetCompactLayout": (OSStatus, [Ref, _P(UInt32), _P(UInt32)]),
    "LBAudioDetectiveGetCompactBands": (OSStatus, [Ref, _P(UInt32), _P(UInt32)]),
    "LBAudioDetectiveCompactFramesToSubfingerprintsDevice": (OSStatus, [Ref, C.c_void_p, UInt64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "LBAudioDetectiveFramesToSubfingerprintsDevice": (OSStatus, [Ref, C.c_void_p, UInt64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "LBAudioDetectiveStreamNew": (Ref, [Ref]),
    "LBAudioDetectiveStreamDispose": (None, [Ref]),
    "LBAudioDetectiveStreamPush": (OSStatus, [Ref, C.c_void_p, UInt64, _P(UInt32)]),
    "LBAudioDetectiveStreamCopyFingerprint": (Ref, [Ref]),
    "LBAudioDetectiveFingerprintClips": (OSStatus, [Ref, C.c_void_p, UInt64, UInt64, C.c_void_p]),
    "LBAudioDetectiveFingerprintClipsFormat": (OSStatus, [Ref, C.c_void_p, UInt32, UInt64, UInt64, C.c_void_p]),
    "LBAudioDetectiveSetKernelVariant": (OSStatus, [Ref, UInt32]),
    "LBAudioDetectiveSetScratchLimit": (OSStatus, [Ref, UInt64]),
    "LBAudioDetectiveSetStageTiming": (OSStatus, [Ref, UInt32]),
    "LBAudioDetectiveGetStageTimes": (OSStatus, [Ref, _P(Float32), _P(Float32), _P(UInt32)]),
    "LBAudioDetectiveFingerprintClipsDeviceTaps": (OSStatus, [Ref, C.c_void_p, UInt64, UInt64, C.c_void_p, C.c_void_p,
                                                              C.c_void_p, C.c_void_p]),
    "LBAudioDetectivePackSubfingerprint": (None, [C.c_void_p, UInt32, C.c_void_p]),
    "LBAudioDetectiveUnpackSubfingerprint": (None, [C.c_void_p, UInt32, C.c_void_p]),
    "LBAudioDetectiveCorpusNew": (Ref, [UInt32, UInt32, UInt64]),
    "LBAudioDetectiveCorpusNewRagged": (Ref, [UInt32, UInt64, UInt64]),
    "LBAudioDetectiveCorpusAppendRaggedPackedDevice": (OSStatus, [Ref, C.c_void_p, C.c_void_p, UInt64, C.c_void_p]),
    "LBAudioDetectiveCorpusGetSubfingerprintTotal": (UInt64, [Ref]),
    "LBAudioDetectiveSynthRaggedCorpusDevice": (OSStatus, [UInt32, UInt64, UInt64, C.c_void_p, UInt64, UInt32, C.c_void_p,
                                                           C.c_void_p]),
    "LBAudioDetectiveCorpusQuerySharded": (OSStatus, [Ref, Ref, UInt32, UInt64, C.c_void_p, C.c_void_p, _P(SInt64), _P(Float32)]),
    "LBAudioDetectiveCorpusQueryBatchSharded": (OSStatus, [Ref, _P(Ref), UInt32, UInt32, UInt64, C.c_void_p, C.c_void_p,
                                                           _P(SInt64), _P(Float32)]),
    "LBAudioDetectiveCorpusQueryBatchShardedWith": (OSStatus, [Ref, _P(Ref), UInt32, UInt32, UInt64, C.c_void_p, C.c_void_p,
                                                               C.c_void_p, _P(SInt64), _P(Float32)]),
    "LBAudioDetectiveSetExchangeTimeout": (None, [UInt32]),
    "LBAudioDetectiveCorpusSetBoundPruning": (OSStatus, [Ref, UInt32]),
    "LBAudioDetectiveCorpusSetBoundPruningThreshold": (OSStatus, [Ref, Float32]),
    "LBAudioDetectiveCorpusGetBoundPruningThreshold": (Float32, [Ref]),
    "LBAudioDetectiveCorpusShardKeysDevice": (C.c_void_p, [Ref]),
    "LBAudioDetectiveCorpusShardKeysHost": (C.c_void_p, [Ref]),
    "LBAudioDetectiveCommGetUniqueId": (OSStatus, [C.c_void_p]),
    "LBAudioDetectiveCommInitRank": (OSStatus, [_P(C.c_void_p), SInt32, C.c_void_p, SInt32]),
    "LBAudioDetectiveCommDestroy": (OSStatus, [C.c_void_p]),
    "LBAudioDetectiveCommGetInfo": (OSStatus, [C.c_void_p, _P(SInt32), _P(SInt32)]),
    "LBAudioDetectiveCorpusDispose": (None, [Ref]),
    "LBAudioDetectiveCorpusGetCount": (UInt64, [Ref]),
    "LBAudioDetectiveCorpusGetEntryStrideBytes": (UInt32, [Ref]),
    "LBAudioDetectiveCorpusAppendPackedDevice": (OSStatus, [Ref, C.c_void_p, UInt64, C.c_void_p]),
    "LBAudioDetectiveCorpusAppendFingerprint": (OSStatus, [Ref, Ref]),
    "LBAudioDetectiveCorpusQuery": (OSStatus, [Ref, Ref, UInt32, _P(SInt64), _P(Float32)]),
    "LBAudioDetectiveCorpusQueryKeyDevice": (OSStatus, [Ref, Ref, UInt32, UInt64, C.c_void_p, C.c_void_p]),
    "LBAudioDetectiveCorpusDecodeKey": (None, [UInt64, _P(SInt64), _P(Float32)]),
    "LBAudioDetectiveCorpusQueryBatch": (OSStatus, [Ref, _P(Ref), UInt32, UInt32, _P(SInt64), _P(Float32)]),
    "LBAudioDetectiveCorpusQueryBatchKeysDevice": (OSStatus, [Ref, _P(Ref), UInt32, UInt32, UInt64, C.c_void_p, C.c_void_p]),
    "LBAudioDetectiveCorpusScoresDevice": (OSStatus, [Ref, Ref, UInt32, C.c_void_p, C.c_void_p]),
    "LBAudioDetectiveCorpusSetKernelVariant": (OSStatus, [Ref, UInt32]),
    "LBAudioDetectiveCorpusSave": (OSStatus, [Ref, C.c_char_p]),
    "LBAudioDetectiveCorpusLoad": (Ref, [C.c_char_p, UInt64]),
    "LBAudioDetectiveSynthClipsDevice": (OSStatus, [UInt32, UInt64, UInt64, UInt32, UInt32, UInt32, C.c_void_p, C.c_void_p]),
    "LBAudioDetectiveSynthCorpusDevice": (OSStatus, [UInt32, UInt64, UInt64, UInt32, UInt32, C.c_void_p, C.c_void_p]),
    "LBAudioDetectiveDeviceCount": (SInt32, []),
    "LBAudioDetectiveDeviceSet": (OSStatus, [SInt32]),
    "LBAudioDetectiveDeviceMalloc": (OSStatus, [_P(C.c_void_p), UInt64]),
    "LBAudioDetectiveDeviceFree": (OSStatus, [C.c_void_p]),
    "LBAudioDetectiveDeviceCopyIn": (OSStatus, [C.c_void_p, C.c_void_p, UInt64]),
    "LBAudioDetectiveDeviceCopyOut": (OSStatus, [C.c_void_p, C.c_void_p, UInt64]),
    "LBAudioDetectiveDeviceSynchronize": (OSStatus, []),
    "LBAudioDetectiveProbeShaderClock": (OSStatus, [C.c_void_p, UInt32, _P(Float64)]),
    "LBAudioDetectiveVersionString": (C.c_char_p, []),
}

CONSTANTS = {
    "kLBAudioDetectiveArgumentInvalid": OSStatus,
    "kLBAudioDetectiveDefaultWindowSize": UInt32,
    "kLBAudioDetectiveDefaultAnalysisStride": UInt32,
    "kLBAudioDetectiveDefaultNumberOfPitchSteps": UInt32,
    "kLBAudioDetectiveDefaultSubfingerprintLength": UInt32,
    "kLBAudioDetectiveDeviceUnavailable": OSStatus,
    "kLBAudioDetectiveDeviceError": OSStatus,
    "kLBAudioDetectiveUnsupportedFile": OSStatus,
    "kLBAudioDetectiveMemFull": OSStatus,
    "kLBAudioDetectiveCollectiveError": OSStatus,
}

_lib = None


def declared_symbols():
    """Every function / constant name include/lbaudiodetective.h declares (parsed from the header)."""
    import re
    text = open(HEADER).read()
    # the inline Objective-C wrappers at the end of the header are source, not exported symbols
    text = re.sub(r"#ifdef __OBJC__\n/\* Objective-C hosts.*?#endif /\* __OBJC__ \*/", "", text, flags=re.S)
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    funcs = set(re.findall(r"\b(LBAudioDetective\w*)\s*\(", text))
    consts = set(re.findall(r"extern const \w+ (kLBAudioDetective\w+);", text))
    return sorted(funcs), sorted(consts)


def lib():
    """The loaded library with argtypes/restype set.  Raises if it is absent (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP library has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or make -C lbaudiodetective_amd/csrc). "
            "There is no CPU fallback.")
    # PyTorch wheels bundle their own libamdhip64 / libhsa-runtime64 with the same SONAMEs as
    # /opt/rocm's.  If ours were loaded first, torch would later bring a SECOND HIP runtime into the
    # process and whichever initialises second finds no device.  Importing torch first makes our
    # library bind to the runtime torch already loaded (one runtime, shared streams and pointers).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def constant(name: str):
    return CONSTANTS[name].in_dll(lib(), name).value
