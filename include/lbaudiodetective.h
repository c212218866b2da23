/*
 * lbaudiodetective.h -- C ABI of the MI355X-native LBAudioDetective hot path.
 *
 * Part 1 re-declares, name for name and argument for argument, the public C interface of
 * the upstream library so that a caller of the reference links against
 * liblbaudiodetective.so unchanged.  Each declaration cites the upstream declaration it
 * replaces (D.h = LBAudioDetective/LBAudioDetective.h, Fp.h = LBAudioDetectiveFingerprint.h,
 * Fr.h = LBAudioDetectiveFrame.h).  Every function that computes (fingerprinting, Haar,
 * sign extraction, compare) runs HIP kernels on the current device; there is no CPU
 * fallback and the calls fail with kLBAudioDetectiveDeviceUnavailable when no GPU is usable.
 *
 * Part 2 adds what the reference lacks and a GPU needs: PCM-in entry points (Apple's
 * ExtAudioFile does not exist here), batch fingerprinting on device-resident clips, and a
 * device-resident reference-fingerprint corpus with a top-1 query that returns a key a
 * host can all-reduce (max) across GPUs.
 *
 * Threading: fingerprints, frames, streams and corpora are, like upstream's objects, not re-entrant -- one call at a
 * time per object; distinct objects are independent.  One process drives one GPU (the current HIP device at the time
 * of the call).  A DETECTIVE may be called from several threads and on several HIP streams (round 3): the frame-row
 * buffer between its two kernels, the kernels' claim counters, its conversion buffers and its timing events exist
 * once per detective, so the library serialises such calls -- a mutex around the host side of every entry point, and a
 * batch call that arrives on another stream than its predecessor first waits on the device (hipStreamWaitEvent) for
 * the predecessor's last kernel.  Results are correct; the calls do not overlap.  For overlap use one detective per
 * stream (they are cheap: a plan of a few tables).  Exception: while a stream is being captured into a hipGraph
 * nothing is recorded or awaited -- ordering replays against other work of the same detective is the caller's.
 *
 * Plain C, plain pointers and sizes only.  "Device pointer" means memory of the current
 * HIP device (e.g. torch.Tensor.data_ptr()); "stream" is a hipStream_t passed as void*
 * (NULL = the default stream).
 */
#ifndef LBAUDIODETECTIVE_H
#define LBAUDIODETECTIVE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- MacTypes / CoreAudio types the upstream headers get from Apple's SDK ------------- */
#if !defined(__MACTYPES__) && !defined(LBAD_HAVE_MACTYPES)
typedef uint32_t UInt32;
typedef int32_t SInt32;
typedef uint64_t UInt64;
typedef int64_t SInt64;
typedef float Float32;
typedef double Float64;
typedef unsigned char Boolean;
typedef SInt32 OSStatus;
enum { noErr = 0 };
#endif

#if !defined(__CoreAudioTypes_h__) && !defined(LBAD_HAVE_COREAUDIOTYPES)
/* layout-compatible with CoreAudio's AudioStreamBasicDescription */
typedef struct AudioStreamBasicDescription {
    Float64 mSampleRate;
    UInt32 mFormatID;
    UInt32 mFormatFlags;
    UInt32 mBytesPerPacket;
    UInt32 mFramesPerPacket;
    UInt32 mBytesPerFrame;
    UInt32 mChannelsPerFrame;
    UInt32 mBitsPerChannel;
    UInt32 mReserved;
} AudioStreamBasicDescription;
enum {
    kAudioFormatLinearPCM = 0x6C70636D, /* 'lpcm' */
    kAudioFormatFlagIsFloat = 1u << 0,
    kAudioFormatFlagIsPacked = 1u << 3
};
#endif

/* Upstream's two file entry points take NSURL* (D.h:218,235).  The library itself is plain C and takes filesystem
 * paths: a C / C++ host calls LBAudioDetectiveProcessAudioURL / ...CompareAudioURLs with a `const char*` in the
 * NSURL's place; an Objective-C host keeps passing NSURL* -- in Objective-C mode the two names are inline
 * wrappers (below, after the declarations) that hand `[[url path] fileSystemRepresentation]` to the path-taking
 * symbols LBAudioDetectiveProcessAudioPath / ...CompareAudioPaths.  No Apple header is needed for that. */
#ifdef __OBJC__
@class NSURL;
typedef NSURL* LBAudioDetectiveURLRef;
#else
typedef const char* LBAudioDetectiveURLRef;
#endif

/* ---- constants (D.h:14-20; values LBAudioDetective.m:20-26) ---------------------------- */
extern const OSStatus kLBAudioDetectiveArgumentInvalid;           /* D.h:14  (= 1) */
extern const UInt32 kLBAudioDetectiveDefaultWindowSize;           /* D.h:16  (= 2048) */
extern const UInt32 kLBAudioDetectiveDefaultAnalysisStride;       /* D.h:17  (= 64) */
extern const UInt32 kLBAudioDetectiveDefaultNumberOfPitchSteps;   /* D.h:18  (= 32) */
extern const UInt32 kLBAudioDetectiveDefaultSubfingerprintLength; /* D.h:20  (= 200) */
/* additions: status codes of this implementation */
extern const OSStatus kLBAudioDetectiveDeviceUnavailable;         /* 'nogp' */
extern const OSStatus kLBAudioDetectiveDeviceError;               /* 'gper' */
extern const OSStatus kLBAudioDetectiveUnsupportedFile;           /* 'fmt?' */
extern const OSStatus kLBAudioDetectiveMemFull;                   /* -108 (MacErrors.h memFullErr): a host allocation failed */
extern const OSStatus kLBAudioDetectiveCollectiveError;           /* 'rccl': RCCL missing or a collective failed */

typedef struct LBAudioDetective* LBAudioDetectiveRef;                       /* D.h:22 */
typedef struct LBAudioDetectiveFingerprint* LBAudioDetectiveFingerprintRef; /* Fp.h:11 */
typedef struct LBAudioDetectiveFrame* LBAudioDetectiveFrameRef;             /* Fr.h:11 */

/* ======================================================================================
 * Part 1a -- detective (D.h)
 * ==================================================================================== */
LBAudioDetectiveRef LBAudioDetectiveNew(void);                                            /* D.h:41 */
OSStatus LBAudioDetectiveDispose(LBAudioDetectiveRef inDetective);                        /* D.h:49 */
AudioStreamBasicDescription LBAudioDetectiveDefaultProcessingFormat(void);                /* D.h:62 */
Float64 LBAudioDetectiveGetProcessingSampleRate(LBAudioDetectiveRef inDetective);         /* D.h:74 */
UInt32 LBAudioDetectiveGetNumberOfPitchSteps(LBAudioDetectiveRef inDetective);            /* D.h:85 */
UInt32 LBAudioDetectiveGetSubfingerprintLength(LBAudioDetectiveRef inDetective);          /* D.h:96,129 */
UInt32 LBAudioDetectiveGetWindowSize(LBAudioDetectiveRef inDetective);                    /* D.h:107 */
UInt32 LBAudioDetectiveGetAnalysisStride(LBAudioDetectiveRef inDetective);                /* D.h:118 */
OSStatus LBAudioDetectiveSetProcessingSampleRate(LBAudioDetectiveRef inDetective, Float64 inSampleRate);        /* D.h:154 */
OSStatus LBAudioDetectiveSetNumberOfPitchSteps(LBAudioDetectiveRef inDetective, UInt32 inNumberOfPitchSteps);   /* D.h:164 */
OSStatus LBAudioDetectiveSetSubfingerprintLength(LBAudioDetectiveRef inDetective, UInt32 inSubfingerprintLength); /* D.h:174,205 */
/* Unlike LBAudioDetective.m:185-187 (which reports an error for every valid size), this
 * returns noErr for a power of two in [16, 8192] and kLBAudioDetectiveArgumentInvalid,
 * leaving the detective unchanged, for anything else. */
OSStatus LBAudioDetectiveSetWindowSize(LBAudioDetectiveRef inDetective, UInt32 inWindowSize);       /* D.h:184 */
OSStatus LBAudioDetectiveSetAnalysisStride(LBAudioDetectiveRef inDetective, UInt32 inAnalysisStride); /* D.h:194 */
/* File front end replacing ExtAudioFile: CAF ('lpcm' or Apple 'ima4') and RIFF/WAVE (PCM, IEEE
 * float), channels averaged to mono, converted to the processing sample rate by a documented
 * windowed-sinc resampler (Apple's converter is closed source; the container is parsed on the host, payload
 * decode, conversion and everything after it run on the GPU), then fingerprinted.
 * Other payloads return kLBAudioDetectiveUnsupportedFile, a missing file -43 (fnfErr). */
OSStatus LBAudioDetectiveProcessAudioPath(LBAudioDetectiveRef inDetective, const char* inFilePath,
                                          LBAudioDetectiveFingerprintRef* outFingerprint);
OSStatus LBAudioDetectiveCompareAudioPaths(LBAudioDetectiveRef inDetective, const char* inFilePath1, const char* inFilePath2,
                                           UInt32 inComparisonRange, Float32* outMatch);
#ifndef __OBJC__
/* the same two functions under upstream's names (exported symbols; LBAudioDetectiveURLRef is const char* here) */
OSStatus LBAudioDetectiveProcessAudioURL(LBAudioDetectiveRef inDetective, LBAudioDetectiveURLRef inFileURL,
                                         LBAudioDetectiveFingerprintRef* outFingerprint); /* D.h:218 */
OSStatus LBAudioDetectiveCompareAudioURLs(LBAudioDetectiveRef inDetective, LBAudioDetectiveURLRef inFileURL1,
                                          LBAudioDetectiveURLRef inFileURL2, UInt32 inComparisonRange,
                                          Float32* outMatch);                             /* D.h:235 */
#endif

/* Addition: many files in one call -- upstream's tests fingerprint 200 files per test, one call each
 * (LBAudioDetectiveTests.m:57-91).  The files are read and parsed by a pool of up to 16 threads that the library starts
 * on the first batch of four or more files and keeps for the life of the process (shared by all detectives); decode,
 * conversion, the window loop and the Haar / sign stage of ALL files run as one launch chain (one per distinct hop when the files' sample rates
 * differ); the converted samples never leave the device.  outFingerprints[i] is what
 * LBAudioDetectiveProcessAudioURL returns for file i (NULL if it failed); outStatuses (optional) receives every
 * file's status, and the call then returns noErr whenever the batch itself could run; without it the first
 * failing file's status is returned.  LBAudioDetectiveProcessAudioURL / ...CompareAudioURLs are this call with
 * one / two files. */
OSStatus LBAudioDetectiveProcessAudioURLs(LBAudioDetectiveRef inDetective, const char* const* inFilePaths, UInt32 inCount,
                                          LBAudioDetectiveFingerprintRef* outFingerprints, OSStatus* outStatuses);
/* A call of many files runs as a two-slot pipeline (round 4): the files go through in runs (an eighth of the call's bytes,
 * at least 16 MB, at most 512 MB), the host reads and parses run i + 1 into a second pinned block while the device decodes,
 * converts and fingerprints run i, and unpacks run i's results while the device works on run i + 1.  Results do not
 * depend on it.  SetFilePipeline(detective, 0) makes the call take one run at a time again (measurement). */
OSStatus LBAudioDetectiveSetFilePipeline(LBAudioDetectiveRef inDetective, UInt32 inEnabled);
/* Parity aid: the file front end alone, on the device -- the mono samples at the processing rate that the window
 * loop of LBAudioDetectiveProcessAudioURL consumes (payload decode + conversion, k_decode.hip / k_resample.hip),
 * copied to a host buffer the caller releases with LBAudioDetectiveFreeSamples; outFileFrames = the file's length
 * in FILE frames (LBAudioDetective.m:236). */
OSStatus LBAudioDetectiveConvertAudioURL(LBAudioDetectiveRef inDetective, const char* inFilePath, Float32** outSamples,
                                         UInt64* outCount, UInt64* outFileFrames, Float64* outFileSampleRate);

/* ======================================================================================
 * Part 1b -- fingerprint (Fp.h).  Sub-fingerprints cross this API as unpacked Booleans.
 * ==================================================================================== */
LBAudioDetectiveFingerprintRef LBAudioDetectiveFingerprintNew(UInt32 inSubfingerprintLength);            /* Fp.h:27 */
void LBAudioDetectiveFingerprintDispose(LBAudioDetectiveFingerprintRef inFingerprint);                  /* Fp.h:35 */
LBAudioDetectiveFingerprintRef LBAudioDetectiveFingerprintCopy(LBAudioDetectiveFingerprintRef inFingerprint); /* Fp.h:45 */
UInt32 LBAudioDetectiveFingerprintGetSubfingerprintLength(LBAudioDetectiveFingerprintRef inFingerprint);  /* Fp.h:59 */
UInt32 LBAudioDetectiveFingerprintGetNumberOfSubfingerprints(LBAudioDetectiveFingerprintRef inFingerprint); /* Fp.h:70 */
UInt32 LBAudioDetectiveFingerprintGetSubfingerprintAtIndex(LBAudioDetectiveFingerprintRef inFingerprint, UInt32 inIndex,
                                                           Boolean* outSubfingerprint);               /* Fp.h:83 */
Boolean LBAudioDetectiveFingerprintSetSubfingerprintLength(LBAudioDetectiveFingerprintRef inFingerprint,
                                                           UInt32* ioSubfingerprintLength);           /* Fp.h:98 */
void LBAudioDetectiveFingerprintAddSubfingerprint(LBAudioDetectiveFingerprintRef inFingerprint,
                                                  Boolean* inSubfingerprint);                         /* Fp.h:108 */
Boolean LBAudioDetectiveFingerprintEqualToFingerprint(LBAudioDetectiveFingerprintRef inFingerprint1,
                                                      LBAudioDetectiveFingerprintRef inFingerprint2); /* Fp.h:122 */
/* GPU: XOR/popcount compare kernel; returns NaN if the device call fails. */
Float32 LBAudioDetectiveFingerprintCompareToFingerprint(LBAudioDetectiveFingerprintRef inFingerprint1,
                                                        LBAudioDetectiveFingerprintRef inFingerprint2,
                                                        UInt32 inRange);                              /* Fp.h:134 */
Float32 LBAudioDetectiveFingerprintCompareSubfingerprints(LBAudioDetectiveFingerprintRef inFingerprint,
                                                          Boolean* inSubfingerprint1, Boolean* inSubfingerprint2,
                                                          UInt32 inRange);                            /* Fp.h:147 */

/* Wire format of a fingerprint: '0'/'1' per Boolean, sub-fingerprints joined by '+' (the string the
 * upstream test helper builds, LBAudioDetectiveTests.m:22-37).  GetString writes a NUL-terminated
 * string when inCapacity suffices and always returns the length needed (without the NUL);
 * NewFromString returns NULL for malformed input. */
UInt64 LBAudioDetectiveFingerprintGetStringLength(LBAudioDetectiveFingerprintRef inFingerprint);
UInt64 LBAudioDetectiveFingerprintGetString(LBAudioDetectiveFingerprintRef inFingerprint, char* outString, UInt64 inCapacity);
LBAudioDetectiveFingerprintRef LBAudioDetectiveFingerprintNewFromString(const char* inString);

/* ======================================================================================
 * Part 1c -- frame (Fr.h; "internal" upstream but used by its Haar test)
 * ==================================================================================== */
LBAudioDetectiveFrameRef LBAudioDetectiveFrameNew(UInt32 inMaxRowCount);                       /* Fr.h:27 */
void LBAudioDetectiveFrameDispose(LBAudioDetectiveFrameRef inFrame);                           /* Fr.h:35 */
LBAudioDetectiveFrameRef LBAudioDetectiveFrameCopy(LBAudioDetectiveFrameRef inFrame);          /* Fr.h:45 */
UInt32 LBAudioDetectiveFrameGetNumberOfRows(LBAudioDetectiveFrameRef inFrame);                 /* Fr.h:59 */
Float32* LBAudioDetectiveFrameGetRow(LBAudioDetectiveFrameRef inFrame, UInt32 inRowIndex);     /* Fr.h:71 */
Float32 LBAudioDetectiveFrameGetValue(LBAudioDetectiveFrameRef inFrame, UInt32 inRowIndex, UInt32 inColumnIndex); /* Fr.h:83 */
Boolean LBAudioDetectiveFrameFull(LBAudioDetectiveFrameRef inFrame);                           /* Fr.h:93 */
Boolean LBAudioDetectiveFrameSetRow(LBAudioDetectiveFrameRef inFrame, Float32* inRow, UInt32 inRowIndex, UInt32 inCount); /* Fr.h:110 */
void LBAudioDetectiveFrameDecompose(LBAudioDetectiveFrameRef inFrame);                         /* Fr.h:121  (GPU Haar) */
size_t LBAudioDetectiveFrameFingerprintSize(LBAudioDetectiveFrameRef inFrame);                 /* Fr.h:131 */
UInt32 LBAudioDetectiveFrameFingerprintLength(LBAudioDetectiveFrameRef inFrame);               /* Fr.h:141 */
void LBAudioDetectiveFrameExtractFingerprint(LBAudioDetectiveFrameRef inFrame, UInt32 inNumberOfWavelets,
                                             Boolean* outFingerprint);                         /* Fr.h:151 (GPU rank) */
Boolean LBAudioDetectiveFrameEqualToFrame(LBAudioDetectiveFrameRef inFrame1, LBAudioDetectiveFrameRef inFrame2); /* Fr.h:162 */

/* ======================================================================================
 * Part 2 -- additions
 * ==================================================================================== */

/* How the file entry points walk a file whose rate differs from the processing rate.
 * Hop mode 1 (default) is what upstream does (LBAudioDetective.m:236,250,275,287-288): its
 * length and its seek offsets are in FILE frames while every read asks for windowSize frames at the
 * PROCESSING rate, so the window count is (fileFrames - windowSize) / analysisStride and the hop is
 * analysisStride * processingRate / fileRate processing-rate samples (rounded, at least 1).
 * Hop mode 0: analysisStride samples at the processing rate, like the PCM entry points. */
OSStatus LBAudioDetectiveSetFileHopMode(LBAudioDetectiveRef inDetective, UInt32 inMode);
/* NOTE on defaults: since round 2 the file entry points default to hop mode 1 and tail mode 1 (upstream's
 * behaviour; round 1 shipped hop mode 0 / zero-filled tails).  For a file whose rate differs from the processing
 * rate the two give different fingerprints, so a corpus built from FILES with the round-1 defaults must be
 * rebuilt, or queried with LBAudioDetectiveSetFileHopMode(d, 0).  Fingerprints of PCM are unaffected. */
/* Hop mode 1 only -- the windows upstream starts so close to the end of the file that ExtAudioFileRead
 * cannot deliver windowSize frames (the last ~windowSize * fileRate / processingRate file frames):
 *   1 (default) the read delivers nothing: inNumberFrames = 0 makes every band of the row 0.0
 *     (LBAudioDetective.m:382-383,404).  This is the behaviour that reproduces the essay's Fig. 24
 *     (eight lossless `_eql` fixtures within 0.5 points);
 *   2 partial reads, literally: the unread part of the in-place FFT buffer keeps the previous window's
 *     packed spectrum and nRead replaces the window size in the band arithmetic (:275,281,351-355,373-395);
 *   0 the unread part is cleared (not upstream). */
OSStatus LBAudioDetectiveSetFileTailMode(LBAudioDetectiveRef inDetective, UInt32 inMode);
/* Converter model of the file entry points (Apple's is closed source): 0 (default) Kaiser-windowed sinc,
 * 24 zero crossings, cut-off 0.92 Nyquist; 1 short sinc (4 zero crossings, cut-off at Nyquist: leaky);
 * 2 linear interpolation (no anti-alias filter). */
OSStatus LBAudioDetectiveSetResamplerMode(LBAudioDetectiveRef inDetective, UInt32 inMode);
/* Decode a file to mono float32, optionally converted to inSampleRate (0 = keep the file's rate): host code
 * (no detective, no device), the samples LBAudioDetectiveProcessAudioURL's device converter produces.
 * The buffer is owned by the caller and released with LBAudioDetectiveFreeSamples. */
OSStatus LBAudioDetectiveReadAudioURL(const char* inFilePath, Float64 inSampleRate, Float32** outSamples,
                                      UInt64* outCount, Float64* outSampleRate);
OSStatus LBAudioDetectiveReadAudioURLWithResampler(const char* inFilePath, Float64 inSampleRate,
                                                   UInt32 inResamplerMode, Float32** outSamples, UInt64* outCount,
                                                   Float64* outSampleRate);
void LBAudioDetectiveFreeSamples(Float32* inSamples);
/* The file loop of LBAudioDetective.m:241-293 on a file that is already decoded and converted:
 * inClientSamples = the whole file at the processing rate, inFileFrames = its length in FILE frames (what
 * kExtAudioFileProperty_FileLengthFrames reports, :236), inHop = processing-rate samples between window
 * starts.  Honours the file tail mode.  LBAudioDetectiveProcessAudioURL in hop mode 1 is
 * decode + convert + this. */
OSStatus LBAudioDetectiveProcessFileStream(LBAudioDetectiveRef inDetective, const Float32* inClientSamples,
                                           UInt64 inClientCount, UInt64 inFileFrames, UInt32 inHop,
                                           LBAudioDetectiveFingerprintRef* outFingerprint);

/* Number of sub-fingerprints a buffer of inNumberOfSamples yields with the detective's
 * window/stride (framing of LBAudioDetective.m:250-255; 0 when shorter than a window). */
UInt64 LBAudioDetectiveGetSubfingerprintCount(LBAudioDetectiveRef inDetective, UInt64 inNumberOfSamples);

/* Replaces the ExtAudioFile loop of LBAudioDetective.m:224-290: host float32 mono PCM that
 * is already at the processing sample rate -> fingerprint. */
OSStatus LBAudioDetectiveProcessPCM(LBAudioDetectiveRef inDetective, const Float32* inSamples,
                                    UInt64 inNumberOfSamples, LBAudioDetectiveFingerprintRef* outFingerprint);
/* LBAudioDetectiveCompareAudioURLs (LBAudioDetective.m:442-464) on two PCM buffers. */
OSStatus LBAudioDetectiveComparePCM(LBAudioDetectiveRef inDetective, const Float32* inSamples1, UInt64 inCount1,
                                    const Float32* inSamples2, UInt64 inCount2, UInt32 inComparisonRange,
                                    Float32* outMatch);

/* Packed sub-fingerprint: LBAD_PACKED_WORDS little-endian 32-bit words; Boolean b of the
 * sub-fingerprint is bit (b & 31) of word (b >> 5); unused high bits are zero.  The device
 * path supports subfingerprintLength <= 256. */
#define LBAD_PACKED_WORDS 8
#define LBAD_PACKED_BYTES 32
#define LBAD_MAX_SUBFINGERPRINT_LENGTH 256

/* Batch hot path: inClips = device pointer to inNumberOfClips x inSamplesPerClip float32,
 * outPacked = device pointer to inNumberOfClips x count x LBAD_PACKED_BYTES, where
 * count = LBAudioDetectiveGetSubfingerprintCount(d, inSamplesPerClip).  Asynchronous on
 * inStream. */
OSStatus LBAudioDetectiveFingerprintClipsDevice(LBAudioDetectiveRef inDetective, const Float32* inClips,
                                                UInt64 inNumberOfClips, UInt64 inSamplesPerClip,
                                                void* outPacked, void* inStream);
/* Same with integer PCM: inSampleFormat 0 = float32, 1 = int16 (sample / 32768), 2 = int32
 * (sample / 2^31).  The conversion LBAudioDetectiveConvertToFormat (LBAudioDetective.m:413-437) hands
 * to AudioConverter is fused into the kernels' PCM load; int16 input halves the HBM read traffic. */
OSStatus LBAudioDetectiveFingerprintClipsDeviceFormat(LBAudioDetectiveRef inDetective, const void* inClips,
                                                      UInt32 inSampleFormat, UInt64 inNumberOfClips,
                                                      UInt64 inSamplesPerClip, void* outPacked, void* inStream);
/* Same, host buffers in and unpacked Booleans out (count x subfingerprintLength per clip). */
OSStatus LBAudioDetectiveFingerprintClips(LBAudioDetectiveRef inDetective, const Float32* inClips,
                                          UInt64 inNumberOfClips, UInt64 inSamplesPerClip, Boolean* outBooleans);
/* Same with integer PCM in host memory (inSampleFormat as above): int16 halves the bytes that cross
 * PCIe, which is what bounds this entry point. */
OSStatus LBAudioDetectiveFingerprintClipsFormat(LBAudioDetectiveRef inDetective, const void* inClips,
                                                UInt32 inSampleFormat, UInt64 inNumberOfClips,
                                                UInt64 inSamplesPerClip, Boolean* outBooleans);
/* Kernel selection for the batch path: 0 = automatic, 1 = generic kernels (any window size / band count / stride),
 * 2 = specialised kernels only -- ArgumentInvalid when the configuration has no specialised stage-1 kernel:
 *   - stride 64: pruned 1024-point FFT for bands that read only bins 0..21; streaming 2048- / 4096-point kernels that
 *     share the early FFT stages between consecutive windows (clips that start on 8-byte boundaries: an even clip
 *     length or a single clip; <= 32 bands);
 *   - register-resident FFT of 256- to 2048-sample windows for any band table at ANY EVEN stride (float32 input when
 *     the stride is not 64; the span of a workgroup's windows must fit the LDS budget) -- among them the hop of 8
 *     samples the file entry points use for 44.1 kHz material at the defaults;
 *   - register Haar / select for frames of 16, 32 or 64 bands.
 * 3 = like 2 but the register-resident 2048-point kernel instead of the streaming one (measurement). */
OSStatus LBAudioDetectiveSetKernelVariant(LBAudioDetectiveRef inDetective, UInt32 inVariant);
/* Measurement knobs of the generic stage-1 kernel (the LDS-tile sizing sweep of tools/sweep_lds_tiles.py):
 * waves per workgroup (0 = automatic; a value the configuration cannot hold falls back to automatic) and
 * whether the shared per-lane twiddle cache is used (default 1).  Results never depend on them. */
OSStatus LBAudioDetectiveSetKernelTuning(LBAudioDetectiveRef inDetective, UInt32 inWavesPerWorkgroup,
                                         UInt32 inTwiddleCache);
/* HBM the frame-row buffer between the two kernels may take (default 512 MiB; 16 KiB per frame at 32
 * bands).  Batches that need more are processed in several launches. */
OSStatus LBAudioDetectiveSetScratchLimit(LBAudioDetectiveRef inDetective, UInt64 inBytes);
/* Measurement aid: once enabled, every batch call records HIP events on its stream around the two
 * kernels; GetStageTimes waits for the last batch and returns, summed over all batch calls since
 * timing was (re-)enabled, the durations (ms) of stage 1 (windows -> frame rows) and stage 2
 * (Haar + select) and the number of launches of each. */
OSStatus LBAudioDetectiveSetStageTiming(LBAudioDetectiveRef inDetective, UInt32 inEnabled);
OSStatus LBAudioDetectiveGetStageTimes(LBAudioDetectiveRef inDetective, Float32* outStage1Ms, Float32* outStage2Ms,
                                       UInt32* outLaunches);
/* Debug taps for stage-level parity tests: device buffers of
 * clips x count x 128 x bands float32 receiving the frame before / after the Haar; either may be NULL. */
OSStatus LBAudioDetectiveFingerprintClipsDeviceTaps(LBAudioDetectiveRef inDetective, const Float32* inClips,
                                                    UInt64 inNumberOfClips, UInt64 inSamplesPerClip,
                                                    void* outPacked, Float32* outFramesRaw, Float32* outFramesHaar,
                                                    void* inStream);

/* Streaming (the essay's live-recording use): PCM arrives in chunks of any size; whenever the
 * buffered samples complete one or more frames they are fingerprinted and appended, and the partial
 * frame is carried to the next call.  After pushing L samples in total the fingerprint equals
 * LBAudioDetectiveProcessPCM on those L samples.  The detective's settings must not change while a
 * stream is open. */
typedef struct LBAudioDetectiveStream* LBAudioDetectiveStreamRef;
LBAudioDetectiveStreamRef LBAudioDetectiveStreamNew(LBAudioDetectiveRef inDetective);
void LBAudioDetectiveStreamDispose(LBAudioDetectiveStreamRef inStream);
OSStatus LBAudioDetectiveStreamPush(LBAudioDetectiveStreamRef inStream, const Float32* inSamples,
                                    UInt64 inNumberOfSamples, UInt32* outNewSubfingerprints);
LBAudioDetectiveFingerprintRef LBAudioDetectiveStreamCopyFingerprint(LBAudioDetectiveStreamRef inStream);

/* Stage 2 alone: inFrames = device pointer to inNumberOfFrames x 128 x bands float32 frame rows (what
 * LBAudioDetectiveFrameSetRow collects upstream) -> packed sub-fingerprints; outFramesHaar (optional)
 * receives the decomposed frames.  Replaces LBAudioDetectiveSynthesizeFingerprint
 * (LBAudioDetective.m:315-331) for a batch of full frames. */
OSStatus LBAudioDetectiveFramesToSubfingerprintsDevice(LBAudioDetectiveRef inDetective, const Float32* inFrames,
                                                       UInt64 inNumberOfFrames, void* outPacked, Float32* outFramesHaar,
                                                       void* inStream);
/* Round 4: where more than half of the bands are structurally empty (a band whose bin range is empty is +0.0 in every
 * window: 17 of the 32 bands at 44.1 kHz / 1024-sample windows) and only one of bands 0..15 is live, the two kernels
 * exchange COMPACT frames -- 128 rows of ONLY the bands that can be non-zero: the live ones of bands 16..31 in ascending
 * order, then that one band (15 floats per row instead of 32 at 44.1 kHz / 1024; never more than 17, hence
 * LBAD_COMPACT_FRAME_FLOATS as an upper bound for buffers) -- and stage 2 runs a sparse form with the same results.
 * GetCompactLayout: noErr and the live left band (32: none) / the number of columns of the row transform that can be
 * non-zero, or ArgumentInvalid when the configuration has no such layout.  GetCompactBands: which bands a row holds, in
 * the order they are stored (outBands may be NULL; *outCount <= 17): a frame is 128 x *outCount floats.
 * CompactFramesToSubfingerprintsDevice: the sparse stage 2 alone on such frames (tests, fuzzers). */
#define LBAD_COMPACT_FRAME_FLOATS 2176
OSStatus LBAudioDetectiveGetCompactLayout(LBAudioDetectiveRef inDetective, UInt32* outLeftBand, UInt32* outLiveColumns);
OSStatus LBAudioDetectiveGetCompactBands(LBAudioDetectiveRef inDetective, UInt32* outBands, UInt32* outCount);
OSStatus LBAudioDetectiveCompactFramesToSubfingerprintsDevice(LBAudioDetectiveRef inDetective, const Float32* inFrames,
                                                              UInt64 inNumberOfFrames, void* outPacked,
                                                              Float32* outFramesHaar, void* inStream);

/* Boolean <-> packed conversion on the host (no arithmetic). */
void LBAudioDetectivePackSubfingerprint(const Boolean* inBooleans, UInt32 inLength, UInt32* outWords);
void LBAudioDetectiveUnpackSubfingerprint(const UInt32* inWords, UInt32 inLength, Boolean* outBooleans);

/* ---- device-resident reference corpus -------------------------------------------------- */
typedef struct LBAudioDetectiveCorpus* LBAudioDetectiveCorpusRef;

/* Every entry has inSubfingerprintsPerEntry sub-fingerprints of inSubfingerprintLength
 * Booleans; inCapacity entries of HBM are reserved up front. */
LBAudioDetectiveCorpusRef LBAudioDetectiveCorpusNew(UInt32 inSubfingerprintLength, UInt32 inSubfingerprintsPerEntry,
                                                    UInt64 inCapacity);
/* Ragged corpus: every entry has its own number (>= 1) of sub-fingerprints -- what the upstream best-match
 * loop actually compares (LBAudioDetectiveTests.m:57-91: one original against ten sequences, all of different
 * lengths; LBAudioDetectiveFingerprint.m:123-146 swaps the two sides and slides the shorter along the longer).
 * inSubfingerprintLength <= 200.  HBM for inSubfingerprintCapacity sub-fingerprints (32 bytes each) and
 * inEntryCapacity entries is reserved up front.  Every query entry point below accepts such a corpus and a query
 * of ANY number of sub-fingerprints; the scan is one launch (k_sliding.hip). */
LBAudioDetectiveCorpusRef LBAudioDetectiveCorpusNewRagged(UInt32 inSubfingerprintLength, UInt64 inEntryCapacity,
                                                          UInt64 inSubfingerprintCapacity);
/* Append inNumberOfEntries entries to a ragged corpus: inPacked = device pointer to the entries' sub-fingerprints
 * back to back in the packed layout (sum(inCounts) x LBAD_PACKED_BYTES), inCounts = HOST array of the entries'
 * sub-fingerprint counts (each >= 1).  Asynchronous on inStream once the counts are read. */
OSStatus LBAudioDetectiveCorpusAppendRaggedPackedDevice(LBAudioDetectiveCorpusRef inCorpus, const void* inPacked,
                                                        const UInt32* inCounts, UInt64 inNumberOfEntries, void* inStream);
/* sub-fingerprints stored, over all entries */
UInt64 LBAudioDetectiveCorpusGetSubfingerprintTotal(LBAudioDetectiveCorpusRef inCorpus);
void LBAudioDetectiveCorpusDispose(LBAudioDetectiveCorpusRef inCorpus);
UInt64 LBAudioDetectiveCorpusGetCount(LBAudioDetectiveCorpusRef inCorpus);
/* bytes of HBM one entry occupies in the scan layout (ragged corpus: bytes per sub-fingerprint) */
UInt32 LBAudioDetectiveCorpusGetEntryStrideBytes(LBAudioDetectiveCorpusRef inCorpus);
/* Append entries from device memory in the packed batch layout
 * (inNumberOfEntries x perEntry x LBAD_PACKED_BYTES). */
OSStatus LBAudioDetectiveCorpusAppendPackedDevice(LBAudioDetectiveCorpusRef inCorpus, const void* inPacked,
                                                  UInt64 inNumberOfEntries, void* inStream);
/* Append one host fingerprint (exactly perEntry sub-fingerprints; any number >= 1 for a ragged corpus). */
OSStatus LBAudioDetectiveCorpusAppendFingerprint(LBAudioDetectiveCorpusRef inCorpus,
                                                 LBAudioDetectiveFingerprintRef inFingerprint);
/* Best-match loop of LBAudioDetectiveTests.m:57-91 over the corpus: the query is the fixed
 * first argument of LBAudioDetectiveFingerprintCompareToFingerprint, every entry the
 * second; strict '<' from 0.0 so the lowest index wins ties and *outIndex = -1 when
 * nothing scores above 0.  inRange == 0 means the sub-fingerprint length
 * (LBAudioDetective.m:443-445). */
OSStatus LBAudioDetectiveCorpusQuery(LBAudioDetectiveCorpusRef inCorpus, LBAudioDetectiveFingerprintRef inQuery,
                                     UInt32 inRange, SInt64* outIndex, Float32* outScore);
/* Sharded form: scans this GPU's entries (global index = inIndexBase + local) and writes
 * ONE 64-bit key = (float bits of the best score << 32) | (0xFFFFFFFF - global index) to
 * the device pointer outKey (0 if the shard is empty).  max() over ranks of the keys
 * (e.g. an RCCL all-reduce with ncclMax on int64) is the global best match; decode with
 * LBAudioDetectiveCorpusDecodeKey.  Asynchronous on inStream. */
OSStatus LBAudioDetectiveCorpusQueryKeyDevice(LBAudioDetectiveCorpusRef inCorpus, LBAudioDetectiveFingerprintRef inQuery,
                                              UInt32 inRange, UInt64 inIndexBase, void* outKey, void* inStream);
void LBAudioDetectiveCorpusDecodeKey(UInt64 inKey, SInt64* outIndex, Float32* outScore);
/* Sharded query with the exchange step inside the library (one process per GPU, every rank holds a contiguous
 * index range of the corpus and calls this with the same query): scan of this rank's entries, then ONE
 * ncclAllReduce(count = number of queries, ncclUint64, ncclMax) of the keys over RCCL / xGMI on inStream, then the
 * 8-byte read-back; every rank receives the global best match, the lowest global index winning ties
 * (LBAudioDetectiveTests.m:80-83 across shards).  inComm is an ncclComm_t passed as void* -- the caller's own, or
 * one made with LBAudioDetectiveCommInitRank.  RCCL is loaded on first use (librccl.so.1; a copy already in the
 * process is reused).  ArgumentInvalid when inIndexBase + the shard's entry count exceeds 2^32 (the key carries
 * a 32-bit global index).  The call is COLLECTIVE: a rank whose own scan cannot run (that error, a NULL corpus, a
 * failed launch) still takes part in the exchange with empty keys, so the other ranks return their result, and
 * reports its own status afterwards; only a NULL communicator or a count of zero returns without the exchange
 * (nothing is allocated by the call: the key block belongs to the corpus). */
OSStatus LBAudioDetectiveCorpusQuerySharded(LBAudioDetectiveCorpusRef inCorpus, LBAudioDetectiveFingerprintRef inQuery,
                                            UInt32 inRange, UInt64 inIndexBase, void* inComm, void* inStream,
                                            SInt64* outIndex, Float32* outScore);
OSStatus LBAudioDetectiveCorpusQueryBatchSharded(LBAudioDetectiveCorpusRef inCorpus,
                                                 const LBAudioDetectiveFingerprintRef* inQueries, UInt32 inCount,
                                                 UInt32 inRange, UInt64 inIndexBase, void* inComm, void* inStream,
                                                 SInt64* outIndices, Float32* outScores);
/* The same with the exchange step handed in: inAllReduce(inContext, keys, count, stream) must leave in `keys` (a
 * DEVICE array of `count` unsigned 64-bit words, in place) the element-wise MAXIMUM over all ranks, ordered on `stream`
 * -- what ncclAllReduce(keys, keys, count, ncclUint64, ncclMax, comm, stream) does; return noErr or an error status.
 * LBAudioDetectiveCorpusQueryBatchSharded is this function with RCCL's all-reduce.  For hosts that bring their own
 * collective (MPI, a different RCCL build) and for tests that run several "ranks" inside one process.  Batches of more
 * than LBAD_SHARD_KEYS queries run as several exchanges, cut the same way on every rank.  After the exchange has been
 * enqueued the call waits for the stream at most LBAudioDetectiveSetExchangeTimeout milliseconds (default 60 000,
 * 0 = for ever) and returns kLBAudioDetectiveCollectiveError when a peer never joined. */
#define LBAD_SHARD_KEYS 4096
typedef OSStatus (*LBAudioDetectiveAllReduceMaxFn)(void* inContext, UInt64* ioDeviceKeys, UInt32 inCount, void* inStream);
OSStatus LBAudioDetectiveCorpusQueryBatchShardedWith(LBAudioDetectiveCorpusRef inCorpus,
                                                     const LBAudioDetectiveFingerprintRef* inQueries, UInt32 inCount,
                                                     UInt32 inRange, UInt64 inIndexBase,
                                                     LBAudioDetectiveAllReduceMaxFn inAllReduce, void* inContext,
                                                     void* inStream, SInt64* outIndices, Float32* outScores);
void LBAudioDetectiveSetExchangeTimeout(UInt32 inMilliseconds);

/* Ragged corpora, top-1 queries (no per-entry scores asked for): a match of 0.7 or better found anywhere in the scan is
 * published at once, and groups of sliding offsets whose sums can no longer reach it -- an upper bound: every remaining
 * sub-fingerprint ratio counted as 1 -- are not finished.  Exact: nothing that could win or tie is dropped, the result is
 * the full scan's (LBAudioDetectiveTests.m:57-91 keeps the best match only).  On by default; 0 switches it off (every
 * offset of every entry is evaluated, as when scores are requested). */
OSStatus LBAudioDetectiveCorpusSetBoundPruning(LBAudioDetectiveCorpusRef inCorpus, UInt32 inEnabled);
/* The score from which a match is published and bounds the rest of the scan (0 < score <= 1; default 0.7: unrelated
 * fingerprints score 0.5 +- 0.03, LBAudioDetective essay p.43).  The bound and its margin: a group of offsets is given up
 * when its largest sum so far + one point per remaining step < published score * query length * 0.999; with float32 sums of
 * at most 8192 terms the rounding of the sums stays three orders of magnitude inside that margin (k_sliding.hip:
 * kPruneMargin), longer queries are scanned without pruning. */
OSStatus LBAudioDetectiveCorpusSetBoundPruningThreshold(LBAudioDetectiveCorpusRef inCorpus, Float32 inScore);
Float32 LBAudioDetectiveCorpusGetBoundPruningThreshold(LBAudioDetectiveCorpusRef inCorpus);
/* the corpus' own key block of a sharded query (LBAD_SHARD_KEYS words on the device, and its pinned host twin) */
unsigned long long* LBAudioDetectiveCorpusShardKeysDevice(LBAudioDetectiveCorpusRef inCorpus);
unsigned long long* LBAudioDetectiveCorpusShardKeysHost(LBAudioDetectiveCorpusRef inCorpus);
/* Communicator helpers for hosts without an RCCL binding of their own: rank 0 obtains a 128-byte id
 * (LBAD_COMM_UNIQUE_ID_BYTES) and hands it to the other ranks by whatever means the host has (a file, a socket,
 * MPI, torch.distributed ...); then every rank calls InitRank with the current HIP device set.  Thin wrappers
 * of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy. */
#define LBAD_COMM_UNIQUE_ID_BYTES 128
OSStatus LBAudioDetectiveCommGetUniqueId(void* outUniqueId);
OSStatus LBAudioDetectiveCommInitRank(void** outComm, SInt32 inNumberOfRanks, const void* inUniqueId, SInt32 inRank);
OSStatus LBAudioDetectiveCommDestroy(void* inComm);
/* What the communicator itself says (ncclCommCount / ncclCommUserRank): the number of ranks that joined it and this
 * rank's number in it.  Round 6: a multi-rank run can check that the library's communicator -- not only the host's own
 * process group -- spans every rank (bench.py --gpus N refuses a line whose count differs from N and names the rank). */
OSStatus LBAudioDetectiveCommGetInfo(void* inComm, SInt32* outNumberOfRanks, SInt32* outRank);
/* Several queries against one pass over the corpus -- the shape of the reference's own test, Q originals against N
 * candidates (LBAudioDetectiveTests.m:57-91).  Uniform corpus: up to 8 queries share each read of an entry.  Ragged
 * corpus (round 5): queries of ONE length share their passes over the records, four per launch (eight in the scan of short
 * queries -- round 6: queries of up to 12 sub-fingerprints, eight per launch in a kernel of their own: eight queries of 5
 * cost 2.2 x one); a batch of mixed lengths runs one group per length.  Results are those of inCount separate
 * LBAudioDetectiveCorpusQuery calls, bit for bit.  The KeysDevice form writes inCount keys to the device pointer
 * outKeys for a sharded max-reduction. */
OSStatus LBAudioDetectiveCorpusQueryBatch(LBAudioDetectiveCorpusRef inCorpus, const LBAudioDetectiveFingerprintRef* inQueries,
                                          UInt32 inCount, UInt32 inRange, SInt64* outIndices, Float32* outScores);
OSStatus LBAudioDetectiveCorpusQueryBatchKeysDevice(LBAudioDetectiveCorpusRef inCorpus,
                                                    const LBAudioDetectiveFingerprintRef* inQueries, UInt32 inCount,
                                                    UInt32 inRange, UInt64 inIndexBase, void* outKeys, void* inStream);
/* Per-entry scores (debug / parity): device pointer to count float32. */
OSStatus LBAudioDetectiveCorpusScoresDevice(LBAudioDetectiveCorpusRef inCorpus, LBAudioDetectiveFingerprintRef inQuery,
                                            UInt32 inRange, Float32* outScores, void* inStream);
/* Binary corpus file ("LBADCRP1" header + the stored entries' planes; a ragged corpus: "LBADCRP2" header + the
 * entries' sub-fingerprint counts + the records); Load recognises both, reserves max(inCapacity, stored count)
 * entries and, for a ragged corpus, records in proportion. */
OSStatus LBAudioDetectiveCorpusSave(LBAudioDetectiveCorpusRef inCorpus, const char* inPath);
LBAudioDetectiveCorpusRef LBAudioDetectiveCorpusLoad(const char* inPath, UInt64 inCapacity);
/* Kernel selection: 0 = automatic, 1 = generic kernel, 2 = specialised plane kernel.  Ragged corpora only: 3 = always hand
 * the entries of fewer than 16 sub-fingerprints that are not longer than the query to the systolic scan (a second launch;
 * automatic: when their share of the work makes it pay), 4 = never. */
OSStatus LBAudioDetectiveCorpusSetKernelVariant(LBAudioDetectiveCorpusRef inCorpus, UInt32 inVariant);

/* ---- synthetic inputs generated on the device (bench / tests) -------------------------- */
/* Integer-arithmetic generator; bit-identical to oracle/lbad_oracle.c:lbo_synth_clip. */
OSStatus LBAudioDetectiveSynthClipsDevice(UInt32 inSeed, UInt64 inFirstClip, UInt64 inNumberOfClips,
                                          UInt32 inSampleRateHz, UInt32 inSamplesPerClip, UInt32 inStereoSum,
                                          Float32* outClips, void* inStream);
/* Ragged form: entry e = sub-fingerprints [inOffsets[e], inOffsets[e + 1]) of the output (inOffsets: DEVICE
 * array of inNumberOfEntries + 1 uint32, inOffsets[0] = 0); sub-fingerprint s of entry e is lbo_synth_entry's. */
OSStatus LBAudioDetectiveSynthRaggedCorpusDevice(UInt32 inSeed, UInt64 inFirstEntry, UInt64 inNumberOfEntries,
                                                 const UInt32* inOffsets, UInt64 inTotalSubfingerprints,
                                                 UInt32 inSubfingerprintLength, void* outPacked, void* inStream);
/* Packed batch layout (entries x perEntry x LBAD_PACKED_BYTES); matches lbo_synth_entry. */
OSStatus LBAudioDetectiveSynthCorpusDevice(UInt32 inSeed, UInt64 inFirstEntry, UInt64 inNumberOfEntries,
                                           UInt32 inSubfingerprintsPerEntry, UInt32 inSubfingerprintLength,
                                           void* outPacked, void* inStream);

/* ---- minimal device plumbing for hosts without a HIP binding --------------------------- */
SInt32 LBAudioDetectiveDeviceCount(void);
OSStatus LBAudioDetectiveDeviceSet(SInt32 inDevice);   /* the current device of this thread (one process drives one GPU) */
OSStatus LBAudioDetectiveDeviceMalloc(void** outPointer, UInt64 inBytes);
OSStatus LBAudioDetectiveDeviceFree(void* inPointer);
OSStatus LBAudioDetectiveDeviceCopyIn(void* inDevice, const void* inHost, UInt64 inBytes);
OSStatus LBAudioDetectiveDeviceCopyOut(void* inHost, const void* inDevice, UInt64 inBytes);
OSStatus LBAudioDetectiveDeviceSynchronize(void);
/* Measurement aid: shader clock (MHz) averaged over inMicroseconds, read inside a one-wave kernel on inStream
 * (s_memtime against the constant 100 MHz s_memrealtime) -- launched on a side stream it reports the clock the
 * chip runs at UNDER the load of whatever else is executing.  Synchronises inStream. */
OSStatus LBAudioDetectiveProbeShaderClock(void* inStream, UInt32 inMicroseconds, Float64* outMegahertz);
const char* LBAudioDetectiveVersionString(void);

#ifdef __cplusplus
}
#endif

#ifdef __OBJC__
/* Objective-C hosts: upstream's NSURL-taking names (D.h:218,235), source compatible with
 * LBAudioDetectiveTests.m:66-68 and the README snippet.  The path is taken with two message sends through the
 * runtime's objc_msgSend; under ARC the intermediate NSString lives until the call has returned. */
#if defined(__has_include)
#if __has_include(<objc/message.h>)
#include <objc/message.h>
#define LBAD_HAVE_OBJC_MESSAGE_H 1
#endif
#endif
#ifndef LBAD_HAVE_OBJC_MESSAGE_H
#ifdef __cplusplus
extern "C" id objc_msgSend(id, SEL, ...);
#else
extern id objc_msgSend(id, SEL, ...);
#endif
#endif
static inline const char* LBAudioDetectivePathOfURL(NSURL* inURL) {
    id path;
    if (!inURL) return (const char*)0;
    path = ((id (*)(id, SEL))(void (*)(void))objc_msgSend)((id)inURL, @selector(path));
    return path ? ((const char* (*)(id, SEL))(void (*)(void))objc_msgSend)(path, @selector(fileSystemRepresentation)) : (const char*)0;
}
static inline OSStatus LBAudioDetectiveProcessAudioURL(LBAudioDetectiveRef inDetective, NSURL* inFileURL,
                                                       LBAudioDetectiveFingerprintRef* outFingerprint) {   /* D.h:218 */
    return LBAudioDetectiveProcessAudioPath(inDetective, LBAudioDetectivePathOfURL(inFileURL), outFingerprint);
}
static inline OSStatus LBAudioDetectiveCompareAudioURLs(LBAudioDetectiveRef inDetective, NSURL* inFileURL1, NSURL* inFileURL2,
                                                        UInt32 inComparisonRange, Float32* outMatch) {   /* D.h:235 */
    return LBAudioDetectiveCompareAudioPaths(inDetective, LBAudioDetectivePathOfURL(inFileURL1),
                                             LBAudioDetectivePathOfURL(inFileURL2), inComparisonRange, outMatch);
}
#endif /* __OBJC__ */
#endif /* LBAUDIODETECTIVE_H */
