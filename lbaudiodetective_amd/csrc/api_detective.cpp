// api_detective.cpp -- LBAudioDetective* driver: configuration, PCM/file entry points and the
// batch hot path.  Mirrors LBAudioDetective/LBAudioDetective.m; line cites refer to it.
#include "internal.hpp"
#include "audiofile.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>

namespace lbad {

uint64_t subfingerprint_count(uint64_t n_samples, uint32_t window, uint32_t stride) {
    // :250-255.  A buffer shorter than one window wraps to ~2^64/stride upstream; here it is 0.
    if (stride == 0 || n_samples < window) return 0;
    return ((n_samples - window) / stride) / kRowsPerFrame;
}

static bool valid_window(uint32_t w) { return w >= kMinWindow && w <= kMaxWindow && (w & (w - 1)) == 0; }

static void free_plan(Plan& p) {
    if (p.d_tw) (void)hipFree(p.d_tw);
    if (p.d_bands) (void)hipFree(p.d_bands);
    if (p.d_bin_const) (void)hipFree(p.d_bin_const);
    if (p.d_claim) (void)hipFree(p.d_claim);
    p = Plan();
}

// (re)build the device tables when the configuration changed since the last call
OSStatus ensure_plan(LBAudioDetective* d) {
    const double rate = d->format.mSampleRate;
    if (!valid_window(d->window) || d->stride == 0 || d->bands == 0 || d->bands > kMaxBands || !(rate > 0.0) ||
        d->subfp_len == 0 || d->subfp_len > LBAD_MAX_SUBFINGERPRINT_LENGTH ||
        d->subfp_len > kRowsPerFrame * d->bands)
        return kLBAudioDetectiveArgumentInvalid;
    Plan& p = d->plan;
    if (p.valid && p.sample_rate == rate && p.window == d->window && p.stride == d->stride && p.bands == d->bands &&
        p.subfp_len == d->subfp_len)
        return noErr;
    if (!device_ready()) return kLBAudioDetectiveDeviceUnavailable;
    free_plan(p);
    p.sample_rate = rate;
    p.window = d->window;
    p.stride = d->stride;
    p.bands = d->bands;
    p.subfp_len = d->subfp_len;
    p.log2w = 0;
    while ((1u << p.log2w) < p.window) ++p.log2w;
    // Extract is asked for subfp_len wavelets but Add keeps subfp_len Booleans (:321-328,
    // Fingerprint.m:91-94): only the first ceil(subfp_len / 2) ranks survive.
    p.keep = (p.subfp_len + 1) / 2;
    make_band_table(rate, p.window, p.bands, p.table);
    std::vector<float> re, im;
    make_twiddles(p.window, re, im);
    const size_t half = p.window / 2;
    LBAD_HIP(hipMalloc(reinterpret_cast<void**>(&p.d_tw), 2 * half * sizeof(float)));
    LBAD_HIP(hipMemcpy(p.d_tw, re.data(), half * sizeof(float), hipMemcpyHostToDevice));
    LBAD_HIP(hipMemcpy(p.d_tw + half, im.data(), half * sizeof(float), hipMemcpyHostToDevice));
    std::vector<uint32_t> tbl(3 * (size_t)p.bands);
    for (uint32_t b = 0; b < p.bands; ++b) {
        tbl[b] = p.table.lo[b];
        tbl[p.bands + b] = p.table.hi[b];
        const float div = (float)(p.table.indices[b + 1] - p.table.indices[b]);  // :404
        std::memcpy(&tbl[2 * p.bands + b], &div, 4);
    }
    LBAD_HIP(hipMalloc(reinterpret_cast<void**>(&p.d_bands), tbl.size() * sizeof(uint32_t)));
    LBAD_HIP(hipMemcpy(p.d_bands, tbl.data(), tbl.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    p.pruned_ok = rows_pruned_supported(p);
    if (p.pruned_ok) {
        std::vector<float> bc;
        rows_pruned_constants(bc);
        LBAD_HIP(hipMalloc(reinterpret_cast<void**>(&p.d_bin_const), bc.size() * sizeof(float)));
        LBAD_HIP(hipMemcpy(p.d_bin_const, bc.data(), bc.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    p.full_ok = rows_full_supported(p);
    if (p.full_ok) LBAD_HIP(hipMalloc(reinterpret_cast<void**>(&p.d_claim), 8 * sizeof(uint32_t)));
    p.valid = true;
    return noErr;
}

static OSStatus ensure_scratch(LBAudioDetective* d, uint64_t floats) {
    if (d->d_frames_cap >= floats) return noErr;
    if (d->d_frames) (void)hipFree(d->d_frames);
    d->d_frames = nullptr;
    d->d_frames_cap = 0;
    LBAD_HIP(hipMalloc(reinterpret_cast<void**>(&d->d_frames), floats * sizeof(float)));
    d->d_frames_cap = floats;
    return noErr;
}

// The batch hot path: every clip -> frames_per_clip packed sub-fingerprints.
OSStatus fingerprint_clips_device(LBAudioDetective* d, const void* d_pcm_raw, uint32_t fmt, uint64_t n_clips,
                                  uint64_t spc, uint32_t* d_packed, float* d_raw, float* d_haar, hipStream_t stream) {
    if (fmt > 2) return kLBAudioDetectiveArgumentInvalid;
    const size_t elem = fmt == 1 ? 2 : 4;
    const char* d_pcm = static_cast<const char*>(d_pcm_raw);
    OSStatus st = ensure_plan(d);
    if (st != noErr) return st;
    const Plan& p = d->plan;
    const uint64_t per = subfingerprint_count(spc, p.window, p.stride);
    if (per == 0 || n_clips == 0) return noErr;
    if (per > 0xFFFFFFFFull / kRowsPerFrame) return kLBAudioDetectiveArgumentInvalid;
    // variant 0: specialised kernels when the configuration has them; 1: generic kernels; 2: specialised or error
    bool special = p.pruned_ok || p.full_ok;
    if (d->variant == 1) special = false;
    if (d->variant == 2 && !special) return kLBAudioDetectiveArgumentInvalid;
    auto stage1 = [&](const void* pcm_in, uint64_t nc, float* frames_out) -> hipError_t {
        if (special && p.pruned_ok)
            return launch_rows_pruned(p, p.d_bin_const, pcm_in, fmt, nc, spc, (uint32_t)per, frames_out, stream);
        if (special) return launch_rows_full(p, pcm_in, fmt, nc, spc, (uint32_t)per, frames_out, stream);
        return launch_fft_bands(p, pcm_in, fmt, nc, spc, (uint32_t)per, frames_out, stream);
    };
    const bool special2 = d->variant != 1 && haar_select32_supported(p);
    auto stage2 = [&](float* frames_in, uint64_t nf, uint32_t* packed_out, float* haar_out) -> hipError_t {
        return special2 ? launch_haar_select32(p, frames_in, nf, packed_out, haar_out, stream)
                        : launch_haar_select(p, frames_in, nf, packed_out, haar_out, stream);
    };
    const uint64_t frame_floats = (uint64_t)kRowsPerFrame * p.bands;
    if (d_raw) {  // the caller's tap buffer doubles as the inter-kernel scratch
        LBAD_HIP(stage1(d_pcm, n_clips, d_raw));
        LBAD_HIP(stage2(d_raw, n_clips * per, d_packed, d_haar));
        return noErr;
    }
    // the inter-stage buffer (16 KiB per frame at 32 bands) is capped; larger batches walk in chunks
    uint64_t chunk = (d->scratch_limit / sizeof(float)) / (per * frame_floats);
    if (chunk == 0) chunk = 1;
    if (chunk > n_clips) chunk = n_clips;
    st = ensure_scratch(d, chunk * per * frame_floats);
    if (st != noErr) return st;
    auto mark = [&]() -> hipError_t {   // events accumulate over calls until SetStageTiming resets them
        if (!d->timing) return hipSuccess;
        if (d->ev_used == d->ev.size()) {
            hipEvent_t e;
            hipError_t err = hipEventCreate(&e);
            if (err != hipSuccess) return err;
            d->ev.push_back(e);
        }
        return hipEventRecord(d->ev[d->ev_used++], stream);
    };
    for (uint64_t c0 = 0; c0 < n_clips; c0 += chunk) {
        const uint64_t nc = (n_clips - c0) < chunk ? (n_clips - c0) : chunk;
        LBAD_HIP(mark());
        LBAD_HIP(stage1(d_pcm + c0 * spc * elem, nc, d->d_frames));
        LBAD_HIP(mark());
        LBAD_HIP(stage2(d->d_frames, nc * per, d_packed + c0 * per * kPackedWords,
                        d_haar ? d_haar + c0 * per * frame_floats : nullptr));
        LBAD_HIP(mark());
    }
    return noErr;
}

}  // namespace lbad

using lbad::ensure_plan;

extern "C" {

LBAudioDetectiveRef LBAudioDetectiveNew(void) {  // :77-90
    LBAudioDetective* d = new LBAudioDetective();
    d->format = LBAudioDetectiveDefaultProcessingFormat();
    d->subfp_len = kLBAudioDetectiveDefaultSubfingerprintLength;
    d->window = kLBAudioDetectiveDefaultWindowSize;
    d->stride = kLBAudioDetectiveDefaultAnalysisStride;
    d->bands = kLBAudioDetectiveDefaultNumberOfPitchSteps;
    return d;
}

OSStatus LBAudioDetectiveDispose(LBAudioDetectiveRef inDetective) {  // :92-111
    if (inDetective == NULL) return kLBAudioDetectiveArgumentInvalid;
    lbad::free_plan(inDetective->plan);
    if (inDetective->d_frames) (void)hipFree(inDetective->d_frames);
    for (hipEvent_t e : inDetective->ev) (void)hipEventDestroy(e);
    delete inDetective;
    return noErr;
}

AudioStreamBasicDescription LBAudioDetectiveDefaultProcessingFormat(void) {  // :116-131
    AudioStreamBasicDescription f;
    std::memset(&f, 0, sizeof(f));
    const UInt32 bytes = sizeof(Float32);
    f.mFormatID = kAudioFormatLinearPCM;
    f.mFormatFlags = kAudioFormatFlagIsFloat | kAudioFormatFlagIsPacked;
    f.mBitsPerChannel = 8 * bytes;
    f.mFramesPerPacket = 1;
    f.mChannelsPerFrame = 1;
    f.mBytesPerPacket = bytes;
    f.mBytesPerFrame = bytes;
    f.mSampleRate = 5512.0;
    return f;
}

Float64 LBAudioDetectiveGetProcessingSampleRate(LBAudioDetectiveRef d) { return d->format.mSampleRate; }  // :133
UInt32 LBAudioDetectiveGetNumberOfPitchSteps(LBAudioDetectiveRef d) { return d->bands; }                 // :137
UInt32 LBAudioDetectiveGetSubfingerprintLength(LBAudioDetectiveRef d) { return d->subfp_len; }           // :141
UInt32 LBAudioDetectiveGetWindowSize(LBAudioDetectiveRef d) { return d->window; }                        // :145
UInt32 LBAudioDetectiveGetAnalysisStride(LBAudioDetectiveRef d) { return d->stride; }                    // :149

// The upstream setters store whatever they are given and return noErr (:156-201); range checks
// happen when the configuration is used.
OSStatus LBAudioDetectiveSetProcessingSampleRate(LBAudioDetectiveRef d, Float64 inSampleRate) {
    d->format.mSampleRate = inSampleRate;
    return noErr;
}
OSStatus LBAudioDetectiveSetNumberOfPitchSteps(LBAudioDetectiveRef d, UInt32 inNumberOfPitchSteps) {
    d->bands = inNumberOfPitchSteps;
    return noErr;
}
OSStatus LBAudioDetectiveSetSubfingerprintLength(LBAudioDetectiveRef d, UInt32 inSubfingerprintLength) {
    d->subfp_len = inSubfingerprintLength;
    return noErr;
}
OSStatus LBAudioDetectiveSetWindowSize(LBAudioDetectiveRef d, UInt32 inWindowSize) {  // :174-195, status inverted on purpose
    if (!lbad::valid_window(inWindowSize)) return kLBAudioDetectiveArgumentInvalid;
    d->window = inWindowSize;
    return noErr;
}
OSStatus LBAudioDetectiveSetAnalysisStride(LBAudioDetectiveRef d, UInt32 inAnalysisStride) {
    d->stride = inAnalysisStride;
    return noErr;
}

OSStatus LBAudioDetectiveSetKernelVariant(LBAudioDetectiveRef d, UInt32 inVariant) {
    if (inVariant > 2) return kLBAudioDetectiveArgumentInvalid;
    d->variant = inVariant;
    return noErr;
}

OSStatus LBAudioDetectiveSetScratchLimit(LBAudioDetectiveRef d, UInt64 inBytes) {
    if (!d || inBytes == 0) return kLBAudioDetectiveArgumentInvalid;
    d->scratch_limit = inBytes;
    return noErr;
}

OSStatus LBAudioDetectiveSetStageTiming(LBAudioDetectiveRef d, UInt32 inEnabled) {
    if (!d) return kLBAudioDetectiveArgumentInvalid;
    d->timing = inEnabled != 0;
    d->ev_used = 0;
    return noErr;
}

OSStatus LBAudioDetectiveGetStageTimes(LBAudioDetectiveRef d, Float32* outStage1Ms, Float32* outStage2Ms,
                                       UInt32* outLaunches) {
    if (!d || !d->timing || d->ev_used < 3) return kLBAudioDetectiveArgumentInvalid;
    float s1 = 0.0f, s2 = 0.0f;
    LBAD_HIP(hipEventSynchronize(d->ev[d->ev_used - 1]));
    for (size_t i = 0; i + 2 < d->ev_used; i += 3) {
        float a = 0.0f, b = 0.0f;
        LBAD_HIP(hipEventElapsedTime(&a, d->ev[i], d->ev[i + 1]));
        LBAD_HIP(hipEventElapsedTime(&b, d->ev[i + 1], d->ev[i + 2]));
        s1 += a;
        s2 += b;
    }
    if (outStage1Ms) *outStage1Ms = s1;
    if (outStage2Ms) *outStage2Ms = s2;
    if (outLaunches) *outLaunches = (UInt32)(d->ev_used / 3);
    return noErr;
}

UInt64 LBAudioDetectiveGetSubfingerprintCount(LBAudioDetectiveRef d, UInt64 inNumberOfSamples) {
    return lbad::subfingerprint_count(inNumberOfSamples, d->window, d->stride);
}

OSStatus LBAudioDetectiveFingerprintClipsDeviceTaps(LBAudioDetectiveRef d, const Float32* inClips, UInt64 inNumberOfClips,
                                                    UInt64 inSamplesPerClip, void* outPacked, Float32* outFramesRaw,
                                                    Float32* outFramesHaar, void* inStream) {
    if (!d || (!inClips && inNumberOfClips) || (!outPacked && inNumberOfClips)) return kLBAudioDetectiveArgumentInvalid;
    return lbad::fingerprint_clips_device(d, inClips, 0, inNumberOfClips, inSamplesPerClip,
                                          static_cast<uint32_t*>(outPacked), outFramesRaw, outFramesHaar,
                                          static_cast<hipStream_t>(inStream));
}

OSStatus LBAudioDetectiveFramesToSubfingerprintsDevice(LBAudioDetectiveRef d, const Float32* inFrames, UInt64 inNumberOfFrames,
                                                       void* outPacked, Float32* outFramesHaar, void* inStream) {
    if (!d || (!inFrames && inNumberOfFrames) || (!outPacked && inNumberOfFrames)) return kLBAudioDetectiveArgumentInvalid;
    OSStatus st = ensure_plan(d);
    if (st != noErr) return st;
    const lbad::Plan& p = d->plan;
    hipStream_t stream = static_cast<hipStream_t>(inStream);
    float* frames = const_cast<float*>(inFrames);   // the kernels only read them
    if (d->variant != 1 && lbad::haar_select32_supported(p)) {
        LBAD_HIP(lbad::launch_haar_select32(p, frames, inNumberOfFrames, static_cast<uint32_t*>(outPacked), outFramesHaar, stream));
    } else {
        if (d->variant == 2) return kLBAudioDetectiveArgumentInvalid;
        LBAD_HIP(lbad::launch_haar_select(p, frames, inNumberOfFrames, static_cast<uint32_t*>(outPacked), outFramesHaar, stream));
    }
    return noErr;
}

OSStatus LBAudioDetectiveFingerprintClipsDeviceFormat(LBAudioDetectiveRef d, const void* inClips, UInt32 inSampleFormat,
                                                      UInt64 inNumberOfClips, UInt64 inSamplesPerClip, void* outPacked,
                                                      void* inStream) {
    if (!d || (!inClips && inNumberOfClips) || (!outPacked && inNumberOfClips)) return kLBAudioDetectiveArgumentInvalid;
    return lbad::fingerprint_clips_device(d, inClips, inSampleFormat, inNumberOfClips, inSamplesPerClip,
                                          static_cast<uint32_t*>(outPacked), nullptr, nullptr,
                                          static_cast<hipStream_t>(inStream));
}

OSStatus LBAudioDetectiveFingerprintClipsDevice(LBAudioDetectiveRef d, const Float32* inClips, UInt64 inNumberOfClips,
                                                UInt64 inSamplesPerClip, void* outPacked, void* inStream) {
    return LBAudioDetectiveFingerprintClipsDeviceTaps(d, inClips, inNumberOfClips, inSamplesPerClip, outPacked, NULL,
                                                      NULL, inStream);
}

OSStatus LBAudioDetectiveFingerprintClips(LBAudioDetectiveRef d, const Float32* inClips, UInt64 inNumberOfClips,
                                          UInt64 inSamplesPerClip, Boolean* outBooleans) {
    return LBAudioDetectiveFingerprintClipsFormat(d, inClips, 0, inNumberOfClips, inSamplesPerClip, outBooleans);
}

OSStatus LBAudioDetectiveFingerprintClipsFormat(LBAudioDetectiveRef d, const void* inClips, UInt32 inSampleFormat,
                                                UInt64 inNumberOfClips, UInt64 inSamplesPerClip,
                                                Boolean* outBooleans) {
    if (!d || !inClips || !outBooleans || inSampleFormat > 2) return kLBAudioDetectiveArgumentInvalid;
    OSStatus st = ensure_plan(d);
    if (st != noErr) return st;
    const uint64_t per = lbad::subfingerprint_count(inSamplesPerClip, d->window, d->stride);
    if (per == 0 || inNumberOfClips == 0) return noErr;
    const size_t pcm_bytes = (size_t)inNumberOfClips * inSamplesPerClip * (inSampleFormat == 1 ? 2 : 4);
    const size_t n_sub = (size_t)inNumberOfClips * per;
    void* d_pcm = nullptr;
    uint32_t* d_packed = nullptr;
    LBAD_HIP(hipMalloc(&d_pcm, pcm_bytes));
    st = lbad::hip_status(hipMalloc(reinterpret_cast<void**>(&d_packed), n_sub * LBAD_PACKED_BYTES), "hipMalloc", __LINE__);
    std::vector<uint32_t> packed(n_sub * LBAD_PACKED_WORDS);
    if (st == noErr) st = lbad::hip_status(hipMemcpy(d_pcm, inClips, pcm_bytes, hipMemcpyHostToDevice), "copy pcm", __LINE__);
    if (st == noErr)
        st = lbad::fingerprint_clips_device(d, d_pcm, inSampleFormat, inNumberOfClips, inSamplesPerClip, d_packed, nullptr,
                                            nullptr, nullptr);
    if (st == noErr)
        st = lbad::hip_status(hipMemcpy(packed.data(), d_packed, n_sub * LBAD_PACKED_BYTES, hipMemcpyDeviceToHost),
                              "copy packed", __LINE__);
    if (d_packed) (void)hipFree(d_packed);
    (void)hipFree(d_pcm);
    if (st != noErr) return st;
    for (size_t s = 0; s < n_sub; ++s)
        LBAudioDetectiveUnpackSubfingerprint(packed.data() + s * LBAD_PACKED_WORDS, d->subfp_len,
                                             outBooleans + s * d->subfp_len);
    return noErr;
}

OSStatus LBAudioDetectiveProcessPCM(LBAudioDetectiveRef d, const Float32* inSamples, UInt64 inNumberOfSamples,
                                    LBAudioDetectiveFingerprintRef* outFingerprint) {
    if (!d || !outFingerprint || (!inSamples && inNumberOfSamples)) return kLBAudioDetectiveArgumentInvalid;
    OSStatus st = ensure_plan(d);
    if (st != noErr) return st;
    const uint64_t per = lbad::subfingerprint_count(inNumberOfSamples, d->window, d->stride);
    std::vector<Boolean> bools((size_t)per * d->subfp_len);
    if (per) {
        st = LBAudioDetectiveFingerprintClips(d, inSamples, 1, inNumberOfSamples, bools.data());
        if (st != noErr) return st;
    }
    // :297-298,326-328 -- New(0), then the length is fixed when the first sub-fingerprint arrives
    LBAudioDetectiveFingerprintRef fp = LBAudioDetectiveFingerprintNew(0);
    for (uint64_t s = 0; s < per; ++s) {
        UInt32 len = d->subfp_len;
        LBAudioDetectiveFingerprintSetSubfingerprintLength(fp, &len);
        LBAudioDetectiveFingerprintAddSubfingerprint(fp, bools.data() + (size_t)s * d->subfp_len);
    }
    *outFingerprint = fp;
    return noErr;
}

OSStatus LBAudioDetectiveComparePCM(LBAudioDetectiveRef d, const Float32* inSamples1, UInt64 inCount1,
                                    const Float32* inSamples2, UInt64 inCount2, UInt32 inComparisonRange,
                                    Float32* outMatch) {  // :442-464 on PCM
    if (inComparisonRange == 0) inComparisonRange = d->subfp_len;  // :443-445
    LBAudioDetectiveFingerprintRef fp1 = NULL, fp2 = NULL;
    OSStatus st = LBAudioDetectiveProcessPCM(d, inSamples1, inCount1, &fp1);
    st = LBAudioDetectiveProcessPCM(d, inSamples2, inCount2, &fp2);  // only the second status gates the compare (:453-458)
    if (st == noErr && fp1 && fp2)
        *outMatch = LBAudioDetectiveFingerprintCompareToFingerprint(fp1, fp2, inComparisonRange);
    LBAudioDetectiveFingerprintDispose(fp1);
    LBAudioDetectiveFingerprintDispose(fp2);
    return st;
}

static OSStatus read_url(LBAudioDetectiveURLRef inFileURL, std::vector<float>& mono, double& rate) {
#ifdef __OBJC__
    const char* path = [[inFileURL path] fileSystemRepresentation];
#else
    const char* path = inFileURL;
#endif
    const lbad::AudioFileStatus fs = lbad::read_audio_file(path, mono, rate);
    if (fs == lbad::AudioFileStatus::NotFound) return -43;  // fnfErr, what ExtAudioFileOpenURL reports
    if (fs != lbad::AudioFileStatus::Ok) return kLBAudioDetectiveUnsupportedFile;
    return noErr;
}

OSStatus LBAudioDetectiveSetFileHopMode(LBAudioDetectiveRef d, UInt32 inMode) {
    if (!d || inMode > 1) return kLBAudioDetectiveArgumentInvalid;
    d->hop_mode = inMode;
    return noErr;
}

OSStatus LBAudioDetectiveReadAudioURL(LBAudioDetectiveURLRef inFileURL, Float64 inSampleRate, Float32** outSamples,
                                      UInt64* outCount, Float64* outSampleRate) {
    if (!inFileURL || !outSamples || !outCount) return kLBAudioDetectiveArgumentInvalid;
    std::vector<float> mono, conv;
    double rate = 0.0;
    OSStatus st = read_url(inFileURL, mono, rate);
    if (st != noErr) return st;
    const std::vector<float>* src = &mono;
    if (inSampleRate > 0.0 && std::fabs(inSampleRate - rate) > 1e-9 * rate) {
        lbad::resample(mono, rate, inSampleRate, conv);
        src = &conv;
        rate = inSampleRate;
    }
    Float32* buf = static_cast<Float32*>(std::malloc(sizeof(Float32) * (src->size() ? src->size() : 1)));
    if (!buf) return kLBAudioDetectiveArgumentInvalid;
    std::memcpy(buf, src->data(), sizeof(Float32) * src->size());
    *outSamples = buf;
    *outCount = src->size();
    if (outSampleRate) *outSampleRate = rate;
    return noErr;
}

void LBAudioDetectiveFreeSamples(Float32* inSamples) { std::free(inSamples); }

OSStatus LBAudioDetectiveProcessAudioURL(LBAudioDetectiveRef d, LBAudioDetectiveURLRef inFileURL,
                                         LBAudioDetectiveFingerprintRef* outFingerprint) {  // :208-308
    if (!inFileURL) return kLBAudioDetectiveArgumentInvalid;  // :211-214
    std::vector<float> file, mono;
    double file_rate = 0.0;
    OSStatus st = read_url(inFileURL, file, file_rate);
    if (st != noErr) return st;
    const double rate = d->format.mSampleRate;
    if (!(rate > 0.0)) return kLBAudioDetectiveArgumentInvalid;
    // ExtAudioFile converts to the client format (:229); here: decode on the host, then resample
    lbad::resample(file, file_rate, rate, mono);
    if (d->hop_mode == 0 || std::fabs(file_rate - rate) <= 1e-9 * rate)
        return LBAudioDetectiveProcessPCM(d, mono.data(), mono.size(), outFingerprint);

    // hop_mode 1 -- what upstream actually does with a file whose rate differs from the processing
    // rate: the length (:236) and the seek offsets (:287-288) are in FILE frames while each
    // read returns windowSize CLIENT frames, so the hop is analysisStride file frames =
    // analysisStride * rate / file_rate client samples and the window count comes from the file length.
    const uint64_t file_frames = file.size();
    if (d->stride == 0 || file_frames < d->window) return LBAudioDetectiveProcessPCM(d, mono.data(), 0, outFingerprint);
    const uint64_t image_width = (file_frames - d->window) / d->stride;                    // :250
    const uint64_t frames = image_width / lbad::kRowsPerFrame;                              // :255
    uint32_t hop = (uint32_t)std::llround((double)d->stride * rate / file_rate);
    if (hop < 1) hop = 1;
    // windows that start near the end of the file read short upstream and keep stale buffer contents;
    // here the stream is zero-padded instead
    const uint64_t need = frames * lbad::kRowsPerFrame * hop + d->window;
    mono.resize(need > mono.size() ? need : mono.size(), 0.0f);
    const uint32_t saved = d->stride;
    d->stride = hop;
    st = LBAudioDetectiveProcessPCM(d, mono.data(), need, outFingerprint);
    d->stride = saved;
    return st;
}

OSStatus LBAudioDetectiveCompareAudioURLs(LBAudioDetectiveRef d, LBAudioDetectiveURLRef inFileURL1,
                                          LBAudioDetectiveURLRef inFileURL2, UInt32 inComparisonRange,
                                          Float32* outMatch) {  // :442-464
    if (inComparisonRange == 0) inComparisonRange = d->subfp_len;
    LBAudioDetectiveFingerprintRef fp1 = NULL, fp2 = NULL;
    OSStatus st = LBAudioDetectiveProcessAudioURL(d, inFileURL1, &fp1);
    st = LBAudioDetectiveProcessAudioURL(d, inFileURL2, &fp2);
    if (st == noErr && fp1 && fp2)
        *outMatch = LBAudioDetectiveFingerprintCompareToFingerprint(fp1, fp2, inComparisonRange);
    LBAudioDetectiveFingerprintDispose(fp1);
    LBAudioDetectiveFingerprintDispose(fp2);
    return st;
}

// ---- streaming: chunked PCM in, the partial frame is carried across calls -----------------------------
struct LBAudioDetectiveStream {
    LBAudioDetectiveRef detective;
    std::vector<Float32> pending;          // starts at a frame boundary of the stream
    LBAudioDetectiveFingerprintRef fingerprint;
};

LBAudioDetectiveStreamRef LBAudioDetectiveStreamNew(LBAudioDetectiveRef inDetective) {
    if (!inDetective) return NULL;
    LBAudioDetectiveStream* s = new LBAudioDetectiveStream();
    s->detective = inDetective;
    s->fingerprint = LBAudioDetectiveFingerprintNew(0);
    return s;
}

void LBAudioDetectiveStreamDispose(LBAudioDetectiveStreamRef inStream) {
    if (!inStream) return;
    LBAudioDetectiveFingerprintDispose(inStream->fingerprint);
    delete inStream;
}

OSStatus LBAudioDetectiveStreamPush(LBAudioDetectiveStreamRef s, const Float32* inSamples, UInt64 inNumberOfSamples,
                                    UInt32* outNewSubfingerprints) {
    if (outNewSubfingerprints) *outNewSubfingerprints = 0;
    if (!s || (!inSamples && inNumberOfSamples)) return kLBAudioDetectiveArgumentInvalid;
    LBAudioDetective* d = s->detective;
    OSStatus st = ensure_plan(d);
    if (st != noErr) return st;
    s->pending.insert(s->pending.end(), inSamples, inSamples + inNumberOfSamples);
    // same rule as the whole-buffer path (:250-255): frame f exists once (L - W) / stride >= 128 (f + 1)
    const uint64_t ready = lbad::subfingerprint_count(s->pending.size(), d->window, d->stride);
    if (ready == 0) return noErr;
    const uint64_t hop = (uint64_t)lbad::kRowsPerFrame * d->stride;
    const uint64_t use = (uint64_t)d->window + ready * hop;            // yields exactly `ready` frames
    std::vector<Boolean> bools((size_t)ready * d->subfp_len);
    st = LBAudioDetectiveFingerprintClips(d, s->pending.data(), 1, use, bools.data());
    if (st != noErr) return st;
    for (uint64_t f = 0; f < ready; ++f) {
        UInt32 len = d->subfp_len;
        LBAudioDetectiveFingerprintSetSubfingerprintLength(s->fingerprint, &len);
        LBAudioDetectiveFingerprintAddSubfingerprint(s->fingerprint, bools.data() + (size_t)f * d->subfp_len);
    }
    s->pending.erase(s->pending.begin(), s->pending.begin() + (size_t)(ready * hop));
    if (outNewSubfingerprints) *outNewSubfingerprints = (UInt32)ready;
    return noErr;
}

LBAudioDetectiveFingerprintRef LBAudioDetectiveStreamCopyFingerprint(LBAudioDetectiveStreamRef s) {
    return s ? LBAudioDetectiveFingerprintCopy(s->fingerprint) : NULL;
}

// ---- synthetic inputs and device plumbing ------------------------------------------------------
OSStatus LBAudioDetectiveSynthClipsDevice(UInt32 inSeed, UInt64 inFirstClip, UInt64 inNumberOfClips,
                                          UInt32 inSampleRateHz, UInt32 inSamplesPerClip, UInt32 inStereoSum,
                                          Float32* outClips, void* inStream) {
    if (!outClips || inSampleRateHz == 0) return kLBAudioDetectiveArgumentInvalid;
    if (!lbad::device_ready()) return kLBAudioDetectiveDeviceUnavailable;
    LBAD_HIP(lbad::launch_synth_clips(inSeed, inFirstClip, inNumberOfClips, inSampleRateHz, inSamplesPerClip,
                                      inStereoSum, outClips, static_cast<hipStream_t>(inStream)));
    return noErr;
}

OSStatus LBAudioDetectiveSynthCorpusDevice(UInt32 inSeed, UInt64 inFirstEntry, UInt64 inNumberOfEntries,
                                           UInt32 inSubfingerprintsPerEntry, UInt32 inSubfingerprintLength,
                                           void* outPacked, void* inStream) {
    if (!outPacked || inSubfingerprintLength == 0 || inSubfingerprintLength > LBAD_MAX_SUBFINGERPRINT_LENGTH)
        return kLBAudioDetectiveArgumentInvalid;
    if (!lbad::device_ready()) return kLBAudioDetectiveDeviceUnavailable;
    LBAD_HIP(lbad::launch_synth_corpus(inSeed, inFirstEntry, inNumberOfEntries, inSubfingerprintsPerEntry,
                                       inSubfingerprintLength, static_cast<uint32_t*>(outPacked),
                                       static_cast<hipStream_t>(inStream)));
    return noErr;
}

SInt32 LBAudioDetectiveDeviceCount(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
OSStatus LBAudioDetectiveDeviceMalloc(void** outPointer, UInt64 inBytes) {
    if (!outPointer) return kLBAudioDetectiveArgumentInvalid;
    if (!lbad::device_ready()) return kLBAudioDetectiveDeviceUnavailable;
    LBAD_HIP(hipMalloc(outPointer, inBytes));
    return noErr;
}
OSStatus LBAudioDetectiveDeviceFree(void* inPointer) {
    LBAD_HIP(hipFree(inPointer));
    return noErr;
}
OSStatus LBAudioDetectiveDeviceCopyIn(void* inDevice, const void* inHost, UInt64 inBytes) {
    LBAD_HIP(hipMemcpy(inDevice, inHost, inBytes, hipMemcpyHostToDevice));
    return noErr;
}
OSStatus LBAudioDetectiveDeviceCopyOut(void* inHost, const void* inDevice, UInt64 inBytes) {
    LBAD_HIP(hipMemcpy(inHost, inDevice, inBytes, hipMemcpyDeviceToHost));
    return noErr;
}
OSStatus LBAudioDetectiveDeviceSynchronize(void) {
    LBAD_HIP(hipDeviceSynchronize());
    return noErr;
}
const char* LBAudioDetectiveVersionString(void) { return "lbaudiodetective-amd 0.1 (gfx950)"; }

}  // extern "C"
