// api_frame.cpp -- LBAudioDetectiveFrame* (ragged rows on the host like upstream; the Haar
// decomposition and the ranked sign extraction run on the GPU).
// Mirrors LBAudioDetective/LBAudioDetectiveFrame.m; line cites refer to it.
#include "internal.hpp"

#include <mutex>

#include <algorithm>
#include <cstring>

extern "C" {

// No C++ exception leaves the library: allocation failure is a NULL reference, a row that was not set (SetRow returns FALSE),
// or a frame that stays as it was (the signatures are upstream's and carry no status).
LBAudioDetectiveFrameRef LBAudioDetectiveFrameNew(UInt32 inMaxRowCount) {  // :22-31
    LBAudioDetectiveFrame* f = nullptr;
    try {
        f = new LBAudioDetectiveFrame();
        f->max_rows = inMaxRowCount;
        f->rows.resize(inMaxRowCount);
        return f;
    } catch (const std::exception&) {
        delete f;
        return NULL;
    }
}

void LBAudioDetectiveFrameDispose(LBAudioDetectiveFrameRef inFrame) { delete inFrame; }  // :33-44

LBAudioDetectiveFrameRef LBAudioDetectiveFrameCopy(LBAudioDetectiveFrameRef inFrame) {  // :46-63
    LBAudioDetectiveFrame* f = nullptr;
    try {
        f = new LBAudioDetectiveFrame(*inFrame);
        // upstream copies exactly rowLength floats of the first numberOfRows rows
        for (uint32_t r = 0; r < f->max_rows; ++r) {
            if (r < f->n_rows) f->rows[r].resize(f->row_length);
            else f->rows[r].clear();
        }
        return f;
    } catch (const std::exception&) {
        delete f;
        return NULL;
    }
}

UInt32 LBAudioDetectiveFrameGetNumberOfRows(LBAudioDetectiveFrameRef inFrame) { return inFrame->n_rows; }  // :67

Float32* LBAudioDetectiveFrameGetRow(LBAudioDetectiveFrameRef inFrame, UInt32 inRowIndex) {  // :71-73 (interior pointer)
    return inFrame->rows[inRowIndex].data();
}

Float32 LBAudioDetectiveFrameGetValue(LBAudioDetectiveFrameRef inFrame, UInt32 inRowIndex, UInt32 inColumnIndex) {  // :75
    return inFrame->rows[inRowIndex][inColumnIndex];
}

Boolean LBAudioDetectiveFrameFull(LBAudioDetectiveFrameRef inFrame) {  // :79-81
    return inFrame->n_rows >= inFrame->max_rows ? 1 : 0;
}

Boolean LBAudioDetectiveFrameSetRow(LBAudioDetectiveFrameRef inFrame, Float32* inRow, UInt32 inRowIndex,
                                    UInt32 inCount) {  // :86-105
    if (LBAudioDetectiveFrameFull(inFrame)) return 0;
    if (inRowIndex >= inFrame->max_rows) return 0;   // upstream writes past its row table here
    try {
        inFrame->rows[inRowIndex].assign(inRow, inRow + inCount);
    } catch (const std::exception&) {
        return 0;
    }
    inFrame->row_length = inFrame->row_length == 0 ? inCount : std::min(inFrame->row_length, inCount);
    inFrame->n_rows++;
    return 1;
}

size_t LBAudioDetectiveFrameFingerprintSize(LBAudioDetectiveFrameRef inFrame) {  // :155-157
    return (size_t)inFrame->n_rows * inFrame->row_length * 2 * sizeof(Boolean);
}

UInt32 LBAudioDetectiveFrameFingerprintLength(LBAudioDetectiveFrameRef inFrame) {  // :159-161
    return inFrame->n_rows * inFrame->row_length * 2;
}

Boolean LBAudioDetectiveFrameEqualToFrame(LBAudioDetectiveFrameRef a, LBAudioDetectiveFrameRef b) {  // :193-210
    if (a->row_length != b->row_length || a->n_rows != b->n_rows) return 0;
    for (uint32_t r = 0; r < a->n_rows; ++r)
        if (std::memcmp(a->rows[r].data(), b->rows[r].data(), a->row_length * sizeof(Float32)) != 0) return 0;
    return 1;
}

}  // extern "C"

namespace {

// gather the first n_rows x row_length block into one dense matrix
bool dense(const LBAudioDetectiveFrame* f, std::vector<float>& m) {
    m.assign((size_t)f->n_rows * f->row_length, 0.0f);
    for (uint32_t r = 0; r < f->n_rows; ++r) {
        if (f->rows[r].size() < f->row_length) return false;
        std::memcpy(m.data() + (size_t)r * f->row_length, f->rows[r].data(), f->row_length * sizeof(float));
    }
    return true;
}

}  // namespace

extern "C" {

// Device memory of the two computing Frame calls: one buffer per device, created on first use and only growing
// (like the other one-off entry points), behind a mutex -- a caller that decomposes many frames pays two copies
// and one launch per call, no allocation.
namespace {
struct FrameContext {
    std::mutex lock;
    void* d = nullptr;
    size_t cap = 0;
    hipStream_t stream = nullptr;
};
FrameContext g_frame[lbad::kMaxDevices];

// returns the context of the current device with at least `bytes` of device memory, locked by `guard`
FrameContext* frame_context(size_t bytes, std::unique_lock<std::mutex>& guard) {
    const int dev = lbad::current_device();
    if (dev < 0 || dev >= lbad::kMaxDevices) return nullptr;
    FrameContext& c = g_frame[dev];
    guard = std::unique_lock<std::mutex>(c.lock);
    if (!c.stream && lbad::hip_status(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking), "stream", __LINE__) != noErr) return nullptr;
    if (c.cap < bytes) {
        void* nd = nullptr;
        const size_t want = bytes + bytes / 4 + 256;
        if (lbad::hip_status(hipMalloc(&nd, want), "hipMalloc", __LINE__) != noErr) return nullptr;
        if (c.d) (void)hipFree(c.d);
        c.d = nd;
        c.cap = want;
    }
    return &c;
}
}  // namespace

void LBAudioDetectiveFrameDecompose(LBAudioDetectiveFrameRef inFrame) {  // :113-132
    LBAudioDetectiveFrame* f = inFrame;
    const uint32_t rows = f->n_rows, cols = f->row_length;
    if (rows == 0 || cols == 0) return;
    try {
    std::vector<float> m;
    if (!dense(f, m)) return;
    if (!lbad::device_ready()) {
        fprintf(stderr, "lbaudiodetective: no HIP device, LBAudioDetectiveFrameDecompose did nothing\n");
        return;
    }
    const size_t bytes = m.size() * sizeof(float);
    std::unique_lock<std::mutex> guard;
    FrameContext* c = frame_context(2 * bytes, guard);
    if (!c) return;
    float* d = static_cast<float*>(c->d);
    bool ok = lbad::hip_status(hipMemcpyAsync(d, m.data(), bytes, hipMemcpyHostToDevice, c->stream), "copy in", __LINE__) == noErr;
    ok = ok && lbad::hip_status(lbad::launch_haar2d_generic(d, d + m.size(), rows, cols, c->stream), "haar", __LINE__) == noErr;
    ok = ok && lbad::hip_status(hipMemcpyAsync(m.data(), d, bytes, hipMemcpyDeviceToHost, c->stream), "copy out", __LINE__) == noErr;
    ok = ok && lbad::hip_status(hipStreamSynchronize(c->stream), "sync", __LINE__) == noErr;
    if (!ok) return;
    for (uint32_t r = 0; r < rows; ++r)
        std::memcpy(f->rows[r].data(), m.data() + (size_t)r * cols, cols * sizeof(float));
    } catch (const std::exception&) {               // out of host memory: the frame stays as it was
    }
}

void LBAudioDetectiveFrameExtractFingerprint(LBAudioDetectiveFrameRef inFrame, UInt32 inNumberOfWavelets,
                                             Boolean* outFingerprint) {  // :165-191
    LBAudioDetectiveFrame* f = inFrame;
    const uint32_t n = f->n_rows * f->row_length;
    if (n == 0 || inNumberOfWavelets == 0) return;
    try {
    std::vector<float> m;
    if (!dense(f, m)) return;
    if (!lbad::device_ready()) {
        fprintf(stderr, "lbaudiodetective: no HIP device, LBAudioDetectiveFrameExtractFingerprint did nothing\n");
        return;
    }
    const uint32_t nw = std::min(inNumberOfWavelets, n);  // upstream indexes past the array beyond n
    const size_t bytes = m.size() * sizeof(float);
    std::unique_lock<std::mutex> guard;
    FrameContext* c = frame_context(bytes + 2 * (size_t)nw, guard);
    if (!c) return;
    float* d = static_cast<float*>(c->d);
    uint8_t* d_out = reinterpret_cast<uint8_t*>(d) + bytes;
    std::vector<uint8_t> flags((size_t)2 * nw, 0);
    bool ok = lbad::hip_status(hipMemcpyAsync(d, m.data(), bytes, hipMemcpyHostToDevice, c->stream), "copy in", __LINE__) == noErr;
    ok = ok && lbad::hip_status(lbad::launch_extract_generic(d, n, nw, d_out, c->stream), "extract", __LINE__) == noErr;
    ok = ok && lbad::hip_status(hipMemcpyAsync(flags.data(), d_out, flags.size(), hipMemcpyDeviceToHost, c->stream), "copy out", __LINE__) == noErr;
    ok = ok && lbad::hip_status(hipStreamSynchronize(c->stream), "sync", __LINE__) == noErr;
    if (!ok) return;
    // upstream only ever writes TRUE into the caller's (pre-zeroed) buffer
    for (size_t i = 0; i < flags.size(); ++i)
        if (flags[i]) outFingerprint[i] = 1;
    } catch (const std::exception&) {               // out of host memory: nothing is written
    }
}

}  // extern "C"
