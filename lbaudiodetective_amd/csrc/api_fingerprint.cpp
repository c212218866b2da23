// api_fingerprint.cpp -- LBAudioDetectiveFingerprint* (container on the host, compare on the GPU).
// Mirrors the behaviour of LBAudioDetective/LBAudioDetectiveFingerprint.m; line cites refer to it.
#include "internal.hpp"

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

LBAudioDetectiveFingerprintRef LBAudioDetectiveFingerprintNew(UInt32 inSubfingerprintLength) {  // :18-26
    LBAudioDetectiveFingerprint* fp = new LBAudioDetectiveFingerprint();
    fp->length = inSubfingerprintLength;
    return fp;
}

void LBAudioDetectiveFingerprintDispose(LBAudioDetectiveFingerprintRef inFingerprint) {  // :28-39 (NULL tolerated)
    delete inFingerprint;
}

LBAudioDetectiveFingerprintRef LBAudioDetectiveFingerprintCopy(LBAudioDetectiveFingerprintRef inFingerprint) {  // :41-59
    return new LBAudioDetectiveFingerprint(*inFingerprint);
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
    inFingerprint->data.insert(inFingerprint->data.end(), inSubfingerprint, inSubfingerprint + inFingerprint->length);
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
    LBAudioDetectiveFingerprint* fp = new LBAudioDetectiveFingerprint();
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

// One-off compare of two slot-packed fingerprints on the GPU (fp1 = "query", fp2 = one entry).
OSStatus compare_slots_once(const std::vector<uint32_t>& fp1, uint32_t n1, const std::vector<uint32_t>& fp2,
                            uint32_t n2, uint32_t length, uint32_t range, float* out) {
    if (!device_ready()) return kLBAudioDetectiveDeviceUnavailable;
    uint32_t* d = nullptr;
    const size_t w1 = fp1.size(), w2 = fp2.size();
    LBAD_HIP(hipMalloc(reinterpret_cast<void**>(&d), (w1 + w2) * sizeof(uint32_t)));
    OSStatus st = noErr;
    auto fail = [&](hipError_t e, const char* what, int line) {
        st = hip_status(e, what, line);
        return st != noErr;
    };
    uint32_t* d_q = d;
    uint32_t* d_e = d + w1;
    unsigned long long* d_key = nullptr;  // [key (8 bytes)][score (4 bytes)]
    float* d_score = nullptr;
    if (fail(hipMalloc(reinterpret_cast<void**>(&d_key), 16), "hipMalloc", __LINE__)) { (void)hipFree(d); return st; }
    d_score = reinterpret_cast<float*>(d_key + 1);
    do {
        if (fail(hipMemcpy(d_q, fp1.data(), w1 * 4, hipMemcpyHostToDevice), "copy fp1", __LINE__)) break;
        if (fail(hipMemcpy(d_e, fp2.data(), w2 * 4, hipMemcpyHostToDevice), "copy fp2", __LINE__)) break;
        if (fail(hipMemset(d_key, 0, 16), "memset", __LINE__)) break;
        if (fail(launch_compare_slots(d_e, 1, n2, length, d_q, n1, range, 0, d_score, d_key, nullptr),
                 "compare kernel", __LINE__)) break;
        if (fail(hipMemcpy(out, d_score, 4, hipMemcpyDeviceToHost), "copy score", __LINE__)) break;
    } while (0);
    (void)hipFree(d_key);
    (void)hipFree(d);
    return st;
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
    if ((size_t)a->count * lbad::kPackedWords * 4 > 48 * 1024) return NAN;  // query must fit LDS
    std::vector<uint32_t> pa, pb;
    lbad::pack_fingerprint(a, pa);
    lbad::pack_fingerprint(b, pb);
    float r = NAN;
    if (lbad::compare_slots_once(pa, a->count, pb, b->count, a->length, inRange, &r) != noErr) return NAN;
    return r;
}

Float32 LBAudioDetectiveFingerprintCompareSubfingerprints(LBAudioDetectiveFingerprintRef inFingerprint,
                                                          Boolean* inSubfingerprint1, Boolean* inSubfingerprint2,
                                                          UInt32 inRange) {  // :151-176
    const uint32_t len = inFingerprint->length;
    if (len == 0) return 0.0f;
    if (len > LBAD_MAX_SUBFINGERPRINT_LENGTH) return NAN;
    std::vector<uint32_t> pa(lbad::kPackedWords), pb(lbad::kPackedWords);
    LBAudioDetectivePackSubfingerprint(inSubfingerprint1, len, pa.data());
    LBAudioDetectivePackSubfingerprint(inSubfingerprint2, len, pb.data());
    float r = NAN;
    if (lbad::compare_slots_once(pa, 1, pb, 1, len, inRange, &r) != noErr) return NAN;
    return r;
}

}  // extern "C"
