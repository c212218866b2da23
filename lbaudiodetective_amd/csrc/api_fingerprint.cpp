// api_fingerprint.cpp -- LBAudioDetectiveFingerprint* (container on the host, compare on the GPU).
// Mirrors the behaviour of LBAudioDetective/LBAudioDetectiveFingerprint.m; line cites refer to it.
#include "internal.hpp"

#include <mutex>
#include <new>

#include <cmath>
#include <cstring>

extern "C" {

const OSStatus kLBAudioDetectiveArgumentInvalid = 1;             // LBAudioDetective.m:20
const UInt32 kLBAudioDetectiveDefaultWindowSize = 2048;          // LBAudioDetective.m:22
const UInt32 kLBAudioDetectiveDefaultAnalysisStride = 64;        // LBAudioDetective.m:23
const UInt32 kLBAudioDetectiveDefaultNumberOfPitchSteps = 32;    // LBAudioDetective.m:24
const UInt32 kLBAudioDetectiveDefaultSubfingerprintLength = 200; // LBAudioDetective.m:26
const OSStatus kLBAudioDetectiveDeviceUnavailable = 0x6E6F6770;  // 'nogp'
const OSStatus kLBAudioDetectiveDeviceError = 0x67706572;        // 'gper'
const OSStatus kLBAudioDetectiveUnsupportedFile = 0x666D743F;    // 'fmt?'
const OSStatus kLBAudioDetectiveMemFull = -108;                  // MacErrors.h memFullErr

void LBAudioDetectivePackSubfingerprint(const Boolean* inBooleans, UInt32 inLength, UInt32* outWords) {
    for (UInt32 w = 0; w < LBAD_PACKED_WORDS; ++w) outWords[w] = 0;
    const UInt32 n = inLength < LBAD_MAX_SUBFINGERPRINT_LENGTH ? inLength : LBAD_MAX_SUBFINGERPRINT_LENGTH;
    for (UInt32 b = 0; b < n; ++b)
        if (inBooleans[b]) outWords[b >> 5] |= 1u << (b & 31);
}

void LBAudioDetectiveUnpackSubfingerprint(const UInt32* inWords, UInt32 inLength, Boolean* outBooleans) {
    for (UInt32 b = 0; b < inLength; ++b)
        outBooleans[b] = b < LBAD_MAX_SUBFINGERPRINT_LENGTH ? (Boolean)((inWords[b >> 5] >> (b & 31)) & 1u) : 0;
}

// No C++ exception leaves the library: where upstream's malloc would have returned NULL, allocation failure here is a NULL
// reference / an unchanged fingerprint (the signatures are upstream's and have no status to return).
LBAudioDetectiveFingerprintRef LBAudioDetectiveFingerprintNew(UInt32 inSubfingerprintLength) {  // :18-26
    LBAudioDetectiveFingerprint* fp = new (std::nothrow) LBAudioDetectiveFingerprint();
    if (fp) fp->length = inSubfingerprintLength;
    return fp;
}

void LBAudioDetectiveFingerprintDispose(LBAudioDetectiveFingerprintRef inFingerprint) {  // :28-39 (NULL tolerated)
    delete inFingerprint;
}

LBAudioDetectiveFingerprintRef LBAudioDetectiveFingerprintCopy(LBAudioDetectiveFingerprintRef inFingerprint) {  // :41-59
    try {
        return new LBAudioDetectiveFingerprint(*inFingerprint);
    } catch (const std::exception&) {             // (bad_alloc from the copy of the Booleans)
        return NULL;
    }
}

UInt32 LBAudioDetectiveFingerprintGetSubfingerprintLength(LBAudioDetectiveFingerprintRef inFingerprint) {  // :64
    return inFingerprint->length;
}

UInt32 LBAudioDetectiveFingerprintGetNumberOfSubfingerprints(LBAudioDetectiveFingerprintRef inFingerprint) {  // :68
    return inFingerprint->count;
}

UInt32 LBAudioDetectiveFingerprintGetSubfingerprintAtIndex(LBAudioDetectiveFingerprintRef inFingerprint, UInt32 inIndex,
                                                           Boolean* outSubfingerprint) {  // :72-76 (no bounds check upstream)
    if (inIndex >= inFingerprint->count) {   // upstream reads past its table here; give zeros instead
        std::memset(outSubfingerprint, 0, inFingerprint->length);
        return inFingerprint->length;
    }
    std::memcpy(outSubfingerprint, inFingerprint->data.data() + (size_t)inIndex * inFingerprint->length,
                inFingerprint->length);
    return inFingerprint->length;
}

Boolean LBAudioDetectiveFingerprintSetSubfingerprintLength(LBAudioDetectiveFingerprintRef inFingerprint,
                                                           UInt32* ioSubfingerprintLength) {  // :81-89
    if (inFingerprint->count > 0) {
        *ioSubfingerprintLength = inFingerprint->length;
        return 0;
    }
    inFingerprint->length = *ioSubfingerprintLength;
    return 1;
}

void LBAudioDetectiveFingerprintAddSubfingerprint(LBAudioDetectiveFingerprintRef inFingerprint,
                                                  Boolean* inSubfingerprint) {  // :91-100 (deep copy of `length` bytes)
    try {
        inFingerprint->data.insert(inFingerprint->data.end(), inSubfingerprint, inSubfingerprint + inFingerprint->length);
    } catch (const std::exception&) {             // out of memory: the fingerprint stays as it was (vector::insert at the end
        return;                                   // of a vector of bytes gives the strong guarantee)
    }
    inFingerprint->count++;
}

Boolean LBAudioDetectiveFingerprintEqualToFingerprint(LBAudioDetectiveFingerprintRef a,
                                                      LBAudioDetectiveFingerprintRef b) {  // :105-117
    if (a->count != b->count || a->length != b->length) return 0;
    return std::memcmp(a->data.data(), b->data.data(), a->data.size()) == 0 ? 1 : 0;
}

// ---- wire format: '0'/'1' per Boolean, sub-fingerprints joined by '+' (the string the upstream test
//      helper builds, LBAudioDetectiveTests.m:22-37, and the essay's client posts to its server) --------
UInt64 LBAudioDetectiveFingerprintGetStringLength(LBAudioDetectiveFingerprintRef fp) {
    if (!fp || fp->count == 0) return 0;
    return (UInt64)fp->count * fp->length + (fp->count - 1);
}

UInt64 LBAudioDetectiveFingerprintGetString(LBAudioDetectiveFingerprintRef fp, char* outString, UInt64 inCapacity) {
    const UInt64 need = LBAudioDetectiveFingerprintGetStringLength(fp);
    if (!outString || inCapacity < need + 1) return need;
    UInt64 at = 0;
    for (uint32_t s = 0; s < fp->count; ++s) {
        if (s) outString[at++] = '+';
        for (uint32_t b = 0; b < fp->length; ++b)
            outString[at++] = fp->data[(size_t)s * fp->length + b] ? '1' : '0';
    }
    outString[at] = 0;
    return need;
}

LBAudioDetectiveFingerprintRef LBAudioDetectiveFingerprintNewFromString(const char* inString) {
    if (!inString) return NULL;
    LBAudioDetectiveFingerprint* fp = new (std::nothrow) LBAudioDetectiveFingerprint();
    if (!fp) return NULL;
    try {
    std::vector<Boolean> row;
    auto flush = [&]() -> bool {
        if (fp->count == 0) fp->length = (uint32_t)row.size();
        if (row.size() != fp->length || row.empty()) return false;
        fp->data.insert(fp->data.end(), row.begin(), row.end());
        fp->count++;
        row.clear();
        return true;
    };
    if (*inString == 0) return fp;   // empty fingerprint
    for (const char* c = inString;; ++c) {
        if (*c == '0' || *c == '1') row.push_back(*c == '1');
        else if (*c == '+' || *c == 0) {
            if (!flush()) { delete fp; return NULL; }
            if (*c == 0) break;
        } else { delete fp; return NULL; }
    }
    return fp;
    } catch (const std::exception&) {
        delete fp;
        return NULL;
    }
}

}  // extern "C"

namespace lbad {

// Pack a host fingerprint into slot words (count * 8).
void pack_fingerprint(const LBAudioDetectiveFingerprint* fp, std::vector<uint32_t>& out) {
    out.assign((size_t)fp->count * kPackedWords, 0u);
    for (uint32_t s = 0; s < fp->count; ++s)
        LBAudioDetectivePackSubfingerprint(fp->data.data() + (size_t)s * fp->length, fp->length,
                                           out.data() + (size_t)s * kPackedWords);
}

// One-off compare of two slot-packed fingerprints on the GPU.  The device buffer, the pinned result word and
// the stream belong to the process (one set per device, created on first use, only growing): a caller that
// compares many fingerprint pairs pays two small copies and one launch per call, no allocation.
namespace {
struct PairContext {
    std::mutex lock;
    uint32_t* d_words = nullptr;
    size_t cap_words = 0;
    unsigned int* d_bits = nullptr;
    unsigned int* h_bits = nullptr;     // pinned
    uint32_t* h_words = nullptr;        // pinned staging for small fingerprints
    size_t h_cap_words = 0;
    hipStream_t stream = nullptr;
};
PairContext g_pair[kMaxDevices];
constexpr size_t kPairPinnedWords = 1u << 20;   // 4 MiB
}  // namespace

OSStatus compare_slots_once(const std::vector<uint32_t>& fp1, uint32_t n1, const std::vector<uint32_t>& fp2,
                            uint32_t n2, uint32_t length, uint32_t range, float* out) {
    if (!device_ready()) return kLBAudioDetectiveDeviceUnavailable;
    const int dev = current_device();
    if (dev < 0 || dev >= kMaxDevices) return kLBAudioDetectiveDeviceUnavailable;
    PairContext& c = g_pair[dev];
    std::lock_guard<std::mutex> guard(c.lock);
    if (!c.stream) {
        LBAD_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
        LBAD_HIP(hipMalloc(reinterpret_cast<void**>(&c.d_bits), sizeof(unsigned int)));
        LBAD_HIP(hipHostMalloc(reinterpret_cast<void**>(&c.h_bits), sizeof(unsigned int), hipHostMallocDefault));
    }
    // Fp.m:123-131 -- the side with more sub-fingerprints is "1" (its non-zero pairs count as possible hits)
    const bool swap = n1 < n2;
    const std::vector<uint32_t>& a = swap ? fp2 : fp1;
    const std::vector<uint32_t>& b = swap ? fp1 : fp2;
    const uint32_t na = swap ? n2 : n1, nb = swap ? n1 : n2;
    const size_t words = a.size() + b.size();
    if (c.cap_words < words) {
        if (c.d_words) (void)hipFree(c.d_words);
        c.d_words = nullptr;
        c.cap_words = 0;
        LBAD_HIP(hipMalloc(reinterpret_cast<void**>(&c.d_words), (words + words / 2) * sizeof(uint32_t)));
        c.cap_words = words + words / 2;
    }
    if (words <= kPairPinnedWords && c.h_cap_words < words) {
        if (c.h_words) (void)hipHostFree(c.h_words);
        c.h_words = nullptr;
        c.h_cap_words = 0;
        LBAD_HIP(hipHostMalloc(reinterpret_cast<void**>(&c.h_words), (words + words / 2) * sizeof(uint32_t), hipHostMallocDefault));
        c.h_cap_words = words + words / 2;
    }
    if (words <= kPairPinnedWords) {
        std::memcpy(c.h_words, a.data(), a.size() * 4);
        std::memcpy(c.h_words + a.size(), b.data(), b.size() * 4);
        LBAD_HIP(hipMemcpyAsync(c.d_words, c.h_words, words * 4, hipMemcpyHostToDevice, c.stream));
    } else {
        LBAD_HIP(hipMemcpyAsync(c.d_words, a.data(), a.size() * 4, hipMemcpyHostToDevice, c.stream));
        LBAD_HIP(hipMemcpyAsync(c.d_words + a.size(), b.data(), b.size() * 4, hipMemcpyHostToDevice, c.stream));
    }
    LBAD_HIP(hipMemsetAsync(c.d_bits, 0, sizeof(unsigned int), c.stream));
    LBAD_HIP(launch_compare_pair(c.d_words, na, c.d_words + a.size(), nb, length, range, c.d_bits, c.stream));
    LBAD_HIP(hipMemcpyAsync(c.h_bits, c.d_bits, sizeof(unsigned int), hipMemcpyDeviceToHost, c.stream));
    LBAD_HIP(hipStreamSynchronize(c.stream));
    std::memcpy(out, c.h_bits, sizeof(float));
    return noErr;
}

}  // namespace lbad

extern "C" {

Float32 LBAudioDetectiveFingerprintCompareToFingerprint(LBAudioDetectiveFingerprintRef inFingerprint1,
                                                        LBAudioDetectiveFingerprintRef inFingerprint2,
                                                        UInt32 inRange) {  // :119-149
    const LBAudioDetectiveFingerprint* a = inFingerprint1;
    const LBAudioDetectiveFingerprint* b = inFingerprint2;
    // An empty side makes every candidate 0/0 = NaN upstream, which Foundation's MAX(A,B)
    // ((a < b) ? b : a) never selects: the result stays 0.
    if (a->count == 0 || b->count == 0) return 0.0f;
    if (a->length != b->length || a->length == 0 || a->length > LBAD_MAX_SUBFINGERPRINT_LENGTH) return NAN;
    try {
        std::vector<uint32_t> pa, pb;
        lbad::pack_fingerprint(a, pa);
        lbad::pack_fingerprint(b, pb);
        float r = NAN;
        if (lbad::compare_slots_once(pa, a->count, pb, b->count, a->length, inRange, &r) != noErr) return NAN;
        return r;
    } catch (const std::exception&) {             // out of host memory: like every other failure of a Float32 entry point
        return NAN;
    }
}

Float32 LBAudioDetectiveFingerprintCompareSubfingerprints(LBAudioDetectiveFingerprintRef inFingerprint,
                                                          Boolean* inSubfingerprint1, Boolean* inSubfingerprint2,
                                                          UInt32 inRange) {  // :151-176
    const uint32_t len = inFingerprint->length;
    if (len == 0) return 0.0f;
    if (len > LBAD_MAX_SUBFINGERPRINT_LENGTH) return NAN;
    try {
        std::vector<uint32_t> pa(lbad::kPackedWords), pb(lbad::kPackedWords);
        LBAudioDetectivePackSubfingerprint(inSubfingerprint1, len, pa.data());
        LBAudioDetectivePackSubfingerprint(inSubfingerprint2, len, pb.data());
        float r = NAN;
        if (lbad::compare_slots_once(pa, 1, pb, 1, len, inRange, &r) != noErr) return NAN;
        return r;
    } catch (const std::exception&) {
        return NAN;
    }
}

}  // extern "C"
