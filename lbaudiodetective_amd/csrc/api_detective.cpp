// api_detective.cpp -- LBAudioDetective* driver: configuration, PCM/file entry points and the
// batch hot path.  Mirrors LBAudioDetective/LBAudioDetective.m; line cites refer to it.
#include "internal.hpp"

#include <new>
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

// stride-dependent part of a plan: which specialised stage-1 kernels apply, and their tables
static OSStatus plan_kernels(Plan& p) {
    p.pruned_ok = rows_pruned_supported(p);
    if (p.pruned_ok && !p.d_bin_const) {
        std::vector<float> bc;
        rows_pruned_constants(bc);
        LBAD_HIP(hipMalloc(reinterpret_cast<void**>(&p.d_bin_const), bc.size() * sizeof(float)));
        LBAD_HIP(hipMemcpy(p.d_bin_const, bc.data(), bc.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    p.full_ok = rows_full_supported(p);
    p.stream_ok = rows_stream_supported(p);
    p.stream2_ok = rows_stream2_supported(p);
    if ((p.full_ok || p.stream_ok || p.stream2_ok) && !p.d_claim) LBAD_HIP(hipMalloc(reinterpret_cast<void**>(&p.d_claim), 8 * sizeof(uint32_t)));
    return noErr;
}

// (re)build the device tables when the configuration changed since the last call
OSStatus ensure_plan(LBAudioDetective* d) {
    const double rate = d->format.mSampleRate;
    if (!valid_window(d->window) || d->stride == 0 || d->bands == 0 || d->bands > kMaxBands || !(rate > 0.0) ||
        d->subfp_len == 0 || d->subfp_len > LBAD_MAX_SUBFINGERPRINT_LENGTH ||
        d->subfp_len > kRowsPerFrame * d->bands)
        return kLBAudioDetectiveArgumentInvalid;
    Plan& p = d->plan;
    p.tune_waves = d->tune_waves;
    p.tune_cache = d->tune_cache;
    if (p.valid && p.sample_rate == rate && p.window == d->window && p.bands == d->bands && p.subfp_len == d->subfp_len) {
        if (p.stride == d->stride) return noErr;
        p.stride = d->stride;          // the tables do not depend on the hop; the kernel choice does
        p.valid = false;
        OSStatus st = plan_kernels(p);
        if (st != noErr) return st;
        p.valid = true;
        return noErr;
    }
    if (!device_ready()) return kLBAudioDetectiveDeviceUnavailable;
    free_plan(p);
    p.tune_waves = d->tune_waves;
    p.tune_cache = d->tune_cache;
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
    plan_sparse(p);
    // [bands] lo, [bands] hi, [bands] divisor as float bits; then where the band's mean of row w goes inside a frame, as
    // multiplier and offset (w * mult + off): for rows of `bands` floats, and for the compact frame of plan.sparse
    // (off 0xFFFFFFFF: not stored); then the first word of the band's power terms in LDS (BandTable::term_at) and, one word,
    // the end of the last band's
    std::vector<uint32_t> tbl(8 * (size_t)p.bands + 1);
    for (uint32_t b = 0; b < p.bands; ++b) {
        tbl[b] = p.table.lo[b];
        tbl[p.bands + b] = p.table.hi[b];
        const float div = (float)(p.table.indices[b + 1] - p.table.indices[b]);  // :404
        std::memcpy(&tbl[2 * p.bands + b], &div, 4);
        tbl[3 * p.bands + b] = p.bands;
        tbl[4 * p.bands + b] = b;
        uint32_t mult = 0, off = 0xFFFFFFFFu;
        if (p.sparse.ok) {               // rows of the n_stored bands that can be non-zero; an empty band is not stored
            const uint32_t pos = b >= 16 ? p.sparse.pos_right[b - 16] : (b == p.sparse.left ? p.sparse.pos_left : 0xFFu);
            if (pos != 0xFFu) { mult = p.sparse.n_stored; off = pos; }
        }
        tbl[5 * p.bands + b] = mult;
        tbl[6 * p.bands + b] = off;
        tbl[7 * p.bands + b] = p.table.term_at[b];
    }
    tbl[8 * (size_t)p.bands] = p.table.term_end;
    LBAD_HIP(hipMalloc(reinterpret_cast<void**>(&p.d_bands), tbl.size() * sizeof(uint32_t)));
    LBAD_HIP(hipMemcpy(p.d_bands, tbl.data(), tbl.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    OSStatus st = plan_kernels(p);
    if (st != noErr) return st;
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

static hipError_t apply_file_tail(const Plan& p, const FileTail& t, const void* d_pcm, uint64_t rows_total, float* frames,
                                  hipStream_t stream) {
    if (t.mode == 3) return launch_empty_rows_batch(p, t.d_files, t.n_files, t.max_rows, frames, stream);
    const uint64_t rows = t.rows ? t.rows : rows_total;
    if (t.first_short >= rows) return hipSuccess;
    float* file_frames = frames + t.row_begin * p.bands;
    if (t.mode == 1)   // inNumberFrames == 0: empty band loops, 0 / divisor in every band (:382-404): +0, NaN for a zero divisor
        return launch_empty_rows(p, file_frames + t.first_short * p.bands, rows - t.first_short, stream);
    if (t.mode == 2)
        return launch_file_tail(p, static_cast<const float*>(d_pcm) + t.pcm_begin, t.n_client, p.stride, t.first_short,
                                (uint32_t)(rows - t.first_short), t.d_tbl, file_frames, stream);
    return hipSuccess;
}

// Orders a batch call behind its predecessor when the two run on different streams (internal.hpp, "Concurrent use").
struct StreamOrder {
    LBAudioDetective* d;
    hipStream_t stream;
    bool active = false;
    hipError_t begin() {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cap) != hipSuccess) { (void)hipGetLastError(); return hipSuccess; }
        if (cap != hipStreamCaptureStatusNone) return hipSuccess;
        if (!d->done) {
            hipError_t e = hipEventCreateWithFlags(&d->done, hipEventDisableTiming);
            if (e != hipSuccess) return e;
        }
        active = true;
        if (d->done_valid && d->done_stream != stream) return hipStreamWaitEvent(stream, d->done, 0);
        return hipSuccess;
    }
    ~StreamOrder() {
        if (!active) return;
        d->done_valid = hipEventRecord(d->done, stream) == hipSuccess;
        d->done_stream = stream;
    }
};

// The batch hot path: every clip -> frames_per_clip packed sub-fingerprints.
OSStatus fingerprint_clips_device(LBAudioDetective* d, const void* d_pcm_raw, uint32_t fmt, uint64_t n_clips,
                                  uint64_t spc, uint32_t* d_packed, float* d_raw, float* d_haar, hipStream_t stream,
                                  const FileTail* tail, size_t n_tails) {
    if (fmt > 2) return kLBAudioDetectiveArgumentInvalid;
    if (tail && (n_clips != 1 || fmt != 0)) return kLBAudioDetectiveArgumentInvalid;
    LBAD_LOCK(d);
    const size_t elem = fmt == 1 ? 2 : 4;
    const char* d_pcm = static_cast<const char*>(d_pcm_raw);
    OSStatus st = ensure_plan(d);
    if (st != noErr) return st;
    const Plan& p = d->plan;
    const uint64_t per = subfingerprint_count(spc, p.window, p.stride);
    if (per == 0 || n_clips == 0) return noErr;
    if (per > 0xFFFFFFFFull / kRowsPerFrame) return kLBAudioDetectiveArgumentInvalid;
    // variant 0: specialised kernels when the configuration has them; 1: generic kernels; 2: specialised or error
    // the streaming kernels read the clips as aligned sample PAIRS: every clip must start on a pair boundary
    const bool pairs_ok = reinterpret_cast<uintptr_t>(d_pcm_raw) % (2 * elem) == 0 && ((spc & 1) == 0 || n_clips == 1);
    const bool stream_ok = p.stream_ok && pairs_ok;
    const bool stream2_ok = p.stream2_ok && pairs_ok && d->variant != 3;
    const bool full_ok = p.full_ok && rows_full_supported_fmt(p, fmt);
    bool special = p.pruned_ok || full_ok || stream_ok || stream2_ok;
    if (d->variant == 3) {                       // measurement: the non-streaming specialised kernel where both exist
        if (!full_ok && !p.pruned_ok) return kLBAudioDetectiveArgumentInvalid;
    }
    if (d->variant == 1) special = false;
    if (d->variant >= 2 && !special) return kLBAudioDetectiveArgumentInvalid;
    StreamOrder order{d, stream};
    LBAD_HIP(order.begin());
    const bool special2 = d->variant != 1 && haar_select32_supported(p);   // (variants 2 and 3 use it as well)
    // rows of 16 floats between the stages where more than half of the bands are structurally empty (plan.sparse): the
    // pruned stage 1 with the sparse stage 2, no file tails (they rewrite whole rows), no tap of the raw frames; variant 4
    // keeps full rows (measurement)
    const bool compact = special && p.pruned_ok && special2 && p.sparse.ok && !tail && !d_raw && d->variant != 4;
    auto stage1 = [&](const void* pcm_in, uint64_t nc, float* frames_out) -> hipError_t {
        if (special && p.pruned_ok)
            return launch_rows_pruned(p, p.d_bin_const, pcm_in, fmt, nc, spc, (uint32_t)per, frames_out, stream, compact);
        if (special && stream2_ok) return launch_rows_stream2(p, pcm_in, fmt, nc, spc, (uint32_t)per, frames_out, stream);
        if (special && full_ok) return launch_rows_full(p, pcm_in, fmt, nc, spc, (uint32_t)per, frames_out, stream);
        if (special) return launch_rows_stream(p, pcm_in, fmt, nc, spc, (uint32_t)per, frames_out, stream);
        return launch_fft_bands(p, pcm_in, fmt, nc, spc, (uint32_t)per, frames_out, stream);
    };
    auto stage2 = [&](float* frames_in, uint64_t nf, uint32_t* packed_out, float* haar_out) -> hipError_t {
        if (compact && haar_out) {      // the sparse form writes the columns that can be non-zero; the others are +0.0
            hipError_t e = hipMemsetAsync(haar_out, 0, nf * kRowsPerFrame * p.bands * sizeof(float), stream);
            if (e != hipSuccess) return e;
        }
        return special2 ? launch_haar_select32(p, frames_in, nf, packed_out, haar_out, stream, compact)
                        : launch_haar_select(p, frames_in, nf, packed_out, haar_out, stream);
    };
    const uint64_t frame_floats = (uint64_t)kRowsPerFrame * p.bands;
    if (d_raw) {  // the caller's tap buffer doubles as the inter-kernel scratch
        LBAD_HIP(stage1(d_pcm, n_clips, d_raw));
        for (size_t t = 0; tail && t < n_tails; ++t) LBAD_HIP(apply_file_tail(p, tail[t], d_pcm, per * kRowsPerFrame, d_raw, stream));
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
        for (size_t t = 0; tail && t < n_tails; ++t)
            LBAD_HIP(apply_file_tail(p, tail[t], d_pcm, per * kRowsPerFrame, d->d_frames, stream));
        LBAD_HIP(mark());
        LBAD_HIP(stage2(d->d_frames, nc * per, d_packed + c0 * per * kPackedWords,
                        d_haar ? d_haar + c0 * per * frame_floats : nullptr));
        LBAD_HIP(mark());
    }
    return noErr;
}

// ---- the one-off entry points (host buffers in, Booleans out) -----------------------------------
// Device buffers, a pinned staging area and a stream live in the detective and only grow, so a
// caller that fingerprints many short buffers (ProcessPCM, StreamPush, ProcessAudioURL) pays no
// allocation per call.
constexpr size_t kPinnedLimit = 512u << 10;  // larger transfers go straight from the caller's memory (measured: staging 1.6 MB costs more than it saves)

OSStatus grow_device(void** ptr, size_t* cap, size_t bytes) {
    if (*cap >= bytes) return noErr;
    if (*ptr) (void)hipFree(*ptr);
    *ptr = nullptr;
    *cap = 0;
    const size_t want = bytes + bytes / 4;
    LBAD_HIP(hipMalloc(ptr, want));
    *cap = want;
    return noErr;
}

static OSStatus ensure_io(LBAudioDetective* d, size_t pcm_bytes, size_t packed_bytes, size_t extra_bytes) {
    OSStatus st = grow_device(&d->d_io_pcm, &d->d_io_pcm_cap, pcm_bytes + extra_bytes + 256);
    if (st != noErr) return st;
    st = grow_device(reinterpret_cast<void**>(&d->d_io_packed), &d->d_io_packed_cap, packed_bytes);
    if (st != noErr) return st;
    if (!d->io_stream) LBAD_HIP(hipStreamCreateWithFlags(&d->io_stream, hipStreamNonBlocking));
    const size_t stage = pcm_bytes + packed_bytes + extra_bytes;
    if (stage <= kPinnedLimit && d->h_io_cap < stage) {
        if (d->h_io) (void)hipHostFree(d->h_io);
        d->h_io = nullptr;
        d->h_io_cap = 0;
        const size_t want = stage + stage / 4;
        LBAD_HIP(hipHostMalloc(&d->h_io, want, hipHostMallocDefault));
        d->h_io_cap = want;
    }
    return noErr;
}

// host clips -> Booleans through the persistent buffers.  `tail`/`tbl_words` describe the end-of-file
// treatment of a single float32 clip (tbl = mode-2 table, copied behind the PCM).
static OSStatus fingerprint_clips_host(LBAudioDetective* d, const void* clips, uint32_t fmt, uint64_t n_clips, uint64_t spc,
                                       Boolean* out, FileTail* tail = nullptr, const std::vector<uint32_t>* tbl = nullptr) {
    OSStatus st = ensure_plan(d);
    if (st != noErr) return st;
    const uint64_t per = subfingerprint_count(spc, d->window, d->stride);
    if (per == 0 || n_clips == 0) return noErr;
    const size_t pcm_bytes = (size_t)n_clips * spc * (fmt == 1 ? 2 : 4);
    const size_t pcm_pad = (pcm_bytes + 255) & ~(size_t)255;
    const size_t n_sub = (size_t)n_clips * per;
    const size_t packed_bytes = n_sub * LBAD_PACKED_BYTES;
    const size_t tbl_bytes = tbl ? tbl->size() * sizeof(uint32_t) : 0;
    st = ensure_io(d, pcm_pad, packed_bytes, tbl_bytes);
    if (st != noErr) return st;
    hipStream_t stream = d->io_stream;
    char* dev = static_cast<char*>(d->d_io_pcm);
    const bool pinned = pcm_pad + packed_bytes + tbl_bytes <= kPinnedLimit;
    char* stage = static_cast<char*>(d->h_io);
    std::vector<uint32_t> packed_big;
    uint32_t* packed_host;
    if (pinned) {
        std::memcpy(stage, clips, pcm_bytes);
        if (tbl_bytes) std::memcpy(stage + pcm_pad, tbl->data(), tbl_bytes);
        LBAD_HIP(hipMemcpyAsync(dev, stage, pcm_pad + tbl_bytes, hipMemcpyHostToDevice, stream));
        packed_host = reinterpret_cast<uint32_t*>(stage + pcm_pad + tbl_bytes);
    } else {
        LBAD_HIP(hipMemcpyAsync(dev, clips, pcm_bytes, hipMemcpyHostToDevice, stream));
        if (tbl_bytes) LBAD_HIP(hipMemcpyAsync(dev + pcm_pad, tbl->data(), tbl_bytes, hipMemcpyHostToDevice, stream));
        packed_big.resize(n_sub * LBAD_PACKED_WORDS);
        packed_host = packed_big.data();
    }
    if (tail) tail->d_tbl = reinterpret_cast<const uint32_t*>(dev + pcm_pad);
    st = fingerprint_clips_device(d, dev, fmt, n_clips, spc, d->d_io_packed, nullptr, nullptr, stream, tail, tail ? 1 : 0);
    if (st != noErr) return st;
    LBAD_HIP(hipMemcpyAsync(packed_host, d->d_io_packed, packed_bytes, hipMemcpyDeviceToHost, stream));
    LBAD_HIP(hipStreamSynchronize(stream));
    for (size_t s = 0; s < n_sub; ++s)
        LBAudioDetectiveUnpackSubfingerprint(packed_host + s * LBAD_PACKED_WORDS, d->subfp_len, out + s * d->subfp_len);
    return noErr;
}

// the device copy of a rational rate pair's phase table (made once per detective) and its pointers in a descriptor
OSStatus device_phase(LBAudioDetective* d, const PhaseTable* host, hipStream_t stream, FileDesc& f) {
    f.ph_p = f.ph_q = 0;
    if (!host || host->m_span == 0) return noErr;
    const DevPhase* found = nullptr;
    for (const DevPhase& e : d->d_phases)
        if (e.host == host) found = &e;
    if (!found) {
        DevPhase e;
        e.host = host;
        const size_t q = (size_t)host->q;
        OSStatus st = hip_status(hipMalloc(reinterpret_cast<void**>(&e.first), q * 4), "phase table", __LINE__);
        if (st == noErr) st = hip_status(hipMalloc(reinterpret_cast<void**>(&e.count), q * 4), "phase table", __LINE__);
        if (st == noErr) st = hip_status(hipMalloc(reinterpret_cast<void**>(&e.wsum), q * 8), "phase table", __LINE__);
        if (st == noErr) st = hip_status(hipMalloc(reinterpret_cast<void**>(&e.w), host->w.size() * 8), "phase table", __LINE__);
        if (st == noErr) st = hip_status(hipMemcpyAsync(e.first, host->first.data(), q * 4, hipMemcpyHostToDevice, stream), "phase table", __LINE__);
        if (st == noErr) st = hip_status(hipMemcpyAsync(e.count, host->count.data(), q * 4, hipMemcpyHostToDevice, stream), "phase table", __LINE__);
        if (st == noErr) st = hip_status(hipMemcpyAsync(e.wsum, host->wsum.data(), q * 8, hipMemcpyHostToDevice, stream), "phase table", __LINE__);
        if (st == noErr) st = hip_status(hipMemcpyAsync(e.w, host->w.data(), host->w.size() * 8, hipMemcpyHostToDevice, stream), "phase table", __LINE__);
        if (st != noErr) {
            if (e.first) (void)hipFree(e.first);
            if (e.count) (void)hipFree(e.count);
            if (e.wsum) (void)hipFree(e.wsum);
            if (e.w) (void)hipFree(e.w);
            return st;
        }
        d->d_phases.push_back(e);           // (the host table lives for the life of the process: pageable copies may finish late)
        found = &d->d_phases.back();
    }
    f.ph_p = host->p; f.ph_q = host->q; f.ph_m_min = host->m_min; f.ph_m_span = host->m_span;
    f.ph_first = found->first; f.ph_count = found->count; f.ph_wsum = found->wsum; f.ph_w = found->w;
    return noErr;
}

// The file front end on the device, one file, samples back on the host (LBAudioDetectiveConvertAudioURL, the parity
// aid): the payload's bytes go up, the SAME table-driven kernels the batch path uses (decode_batch_kernel,
// resample_batch_kernel, here with one descriptor) decode and convert, the converted samples come back.
static OSStatus convert_file_on_device(LBAudioDetective* d, const AudioPayload& a, double rate_out, uint32_t mode,
                                       std::vector<float>& out) {
    out.clear();
    ResamplePlan rp;
    if (!resample_plan(a.count, a.sample_rate, rate_out, mode, rp)) return kLBAudioDetectiveArgumentInvalid;
    if (a.count == 0) return noErr;
    out.resize(rp.n_out);
    if (rp.n_out == 0) return noErr;
    OSStatus st = grow_device(&d->d_rs_bytes, &d->d_rs_bytes_cap, a.len);
    if (st != noErr) return st;
    st = grow_device(&d->d_rs_in, &d->d_rs_in_cap, a.total_frames * sizeof(float));
    if (st != noErr) return st;
    st = grow_device(&d->d_rs_out, &d->d_rs_out_cap, rp.n_out * sizeof(float));
    if (st != noErr) return st;
    st = grow_device(&d->d_rs_desc, &d->d_rs_desc_cap, sizeof(FileDesc));
    if (st != noErr) return st;
    if (!d->io_stream) LBAD_HIP(hipStreamCreateWithFlags(&d->io_stream, hipStreamNonBlocking));
    hipStream_t stream = d->io_stream;
    FileDesc f;
    std::memset(&f, 0, sizeof(f));
    f.kind = (uint32_t)a.kind; f.channels = a.channels; f.bits = a.bits;
    f.flags = (a.is_float ? 1u : 0u) | (a.little ? 2u : 0u);
    f.total_frames = a.total_frames; f.first = a.first; f.n_in = a.count;
    f.n_write = rp.n_out; f.mode = rp.mode; f.copy = rp.copy ? 1u : 0u;
    f.ratio = rp.ratio; f.scale = rp.scale; f.half = rp.half;
    st = device_phase(d, rp.copy ? nullptr : rp.phases, stream, f);
    if (st != noErr) return st;
    LBAD_HIP(hipMemcpyAsync(d->d_rs_bytes, a.bytes + a.off, a.len, hipMemcpyHostToDevice, stream));
    LBAD_HIP(hipMemcpyAsync(d->d_rs_desc, &f, sizeof(f), hipMemcpyHostToDevice, stream));
    const double* d_table = nullptr;
    uint64_t table_n = 0;
    if (!rp.copy && mode < 2) {
        table_n = rp.table->size();
        if (!d->d_rs_table[mode]) {
            LBAD_HIP(hipMalloc(reinterpret_cast<void**>(&d->d_rs_table[mode]), table_n * sizeof(double)));
            LBAD_HIP(hipMemcpyAsync(d->d_rs_table[mode], rp.table->data(), table_n * sizeof(double), hipMemcpyHostToDevice, stream));
        }
        d_table = d->d_rs_table[mode];
    }
    const FileDesc* d_files = static_cast<const FileDesc*>(d->d_rs_desc);
    const uint64_t units = a.kind == AudioPayload::Ima4 ? a.total_frames / 64 : a.total_frames;
    LBAD_HIP(launch_decode_batch(d_files, 1, units, static_cast<const uint8_t*>(d->d_rs_bytes), static_cast<float*>(d->d_rs_in), stream));
    LBAD_HIP(launch_resample_batch(d_files, 1, rp.n_out, static_cast<const float*>(d->d_rs_in), rp.table_res, d_table, table_n,
                                   static_cast<float*>(d->d_rs_out), stream));
    LBAD_HIP(hipMemcpyAsync(out.data(), d->d_rs_out, rp.n_out * sizeof(float), hipMemcpyDeviceToHost, stream));
    LBAD_HIP(hipStreamSynchronize(stream));
    return noErr;
}

// :297-298,326-328 -- New(0), then the length is fixed when the first sub-fingerprint arrives
LBAudioDetectiveFingerprintRef fingerprint_from_bools(const LBAudioDetective* d, const Boolean* bools, uint64_t per) {
    LBAudioDetectiveFingerprintRef fp = LBAudioDetectiveFingerprintNew(0);
    for (uint64_t s = 0; s < per; ++s) {
        UInt32 len = d->subfp_len;
        LBAudioDetectiveFingerprintSetSubfingerprintLength(fp, &len);
        LBAudioDetectiveFingerprintAddSubfingerprint(fp, const_cast<Boolean*>(bools) + (size_t)s * d->subfp_len);
    }
    return fp;
}

}  // namespace lbad

using lbad::ensure_plan;


extern "C" {

LBAudioDetectiveRef LBAudioDetectiveNew(void) {  // :77-90
    LBAudioDetective* d = new (std::nothrow) LBAudioDetective();      // (no C++ exception leaves the library: NULL like a failed malloc)
    if (!d) return NULL;
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
    if (inDetective->d_io_pcm) (void)hipFree(inDetective->d_io_pcm);
    if (inDetective->d_io_packed) (void)hipFree(inDetective->d_io_packed);
    if (inDetective->h_io) (void)hipHostFree(inDetective->h_io);
    if (inDetective->d_rs_bytes) (void)hipFree(inDetective->d_rs_bytes);
    if (inDetective->d_rs_bytes_b) (void)hipFree(inDetective->d_rs_bytes_b);
    if (inDetective->up_stream) { (void)hipStreamSynchronize(inDetective->up_stream); (void)hipStreamDestroy(inDetective->up_stream); }
    for (hipEvent_t e : inDetective->up_done) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : inDetective->bytes_free) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : inDetective->packed_done) if (e) (void)hipEventDestroy(e);
    if (inDetective->h_files_b) (void)hipHostFree(inDetective->h_files_b);
    if (inDetective->h_packed_b) (void)hipHostFree(inDetective->h_packed_b);
    if (inDetective->d_rs_in) (void)hipFree(inDetective->d_rs_in);
    if (inDetective->d_rs_out) (void)hipFree(inDetective->d_rs_out);
    if (inDetective->d_rs_tail) (void)hipFree(inDetective->d_rs_tail);
    if (inDetective->d_rs_desc) (void)hipFree(inDetective->d_rs_desc);
    if (inDetective->h_files) (void)hipHostFree(inDetective->h_files);
    if (inDetective->h_packed) (void)hipHostFree(inDetective->h_packed);
    for (double* t : inDetective->d_rs_table)
        if (t) (void)hipFree(t);
    for (lbad::DevPhase& e : inDetective->d_phases) {
        (void)hipFree(e.first); (void)hipFree(e.count); (void)hipFree(e.wsum); (void)hipFree(e.w);
    }
    if (inDetective->io_stream) (void)hipStreamDestroy(inDetective->io_stream);
    for (hipEvent_t e : inDetective->ev) (void)hipEventDestroy(e);
    if (inDetective->done) (void)hipEventDestroy(inDetective->done);
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
    LBAD_LOCK(d);
    d->format.mSampleRate = inSampleRate;
    return noErr;
}
OSStatus LBAudioDetectiveSetNumberOfPitchSteps(LBAudioDetectiveRef d, UInt32 inNumberOfPitchSteps) {
    LBAD_LOCK(d);
    d->bands = inNumberOfPitchSteps;
    return noErr;
}
OSStatus LBAudioDetectiveSetSubfingerprintLength(LBAudioDetectiveRef d, UInt32 inSubfingerprintLength) {
    LBAD_LOCK(d);
    d->subfp_len = inSubfingerprintLength;
    return noErr;
}
OSStatus LBAudioDetectiveSetWindowSize(LBAudioDetectiveRef d, UInt32 inWindowSize) {  // :174-195, status inverted on purpose
    LBAD_LOCK(d);
    if (!lbad::valid_window(inWindowSize)) return kLBAudioDetectiveArgumentInvalid;
    d->window = inWindowSize;
    return noErr;
}
OSStatus LBAudioDetectiveSetAnalysisStride(LBAudioDetectiveRef d, UInt32 inAnalysisStride) {
    LBAD_LOCK(d);
    d->stride = inAnalysisStride;
    return noErr;
}

OSStatus LBAudioDetectiveSetKernelVariant(LBAudioDetectiveRef d, UInt32 inVariant) {
    LBAD_LOCK(d);
    if (inVariant > 4) return kLBAudioDetectiveArgumentInvalid;   // 4: variant 2 with full rows between the stages (measurement)
    d->variant = inVariant;
    return noErr;
}

OSStatus LBAudioDetectiveSetKernelTuning(LBAudioDetectiveRef d, UInt32 inWavesPerWorkgroup, UInt32 inTwiddleCache) {
    LBAD_LOCK(d);
    if (!d || inWavesPerWorkgroup > 16) return kLBAudioDetectiveArgumentInvalid;
    d->tune_waves = inWavesPerWorkgroup;
    d->tune_cache = inTwiddleCache != 0;
    return noErr;
}

OSStatus LBAudioDetectiveSetScratchLimit(LBAudioDetectiveRef d, UInt64 inBytes) {
    LBAD_LOCK(d);
    if (!d || inBytes == 0) return kLBAudioDetectiveArgumentInvalid;
    d->scratch_limit = inBytes;
    return noErr;
}

OSStatus LBAudioDetectiveSetStageTiming(LBAudioDetectiveRef d, UInt32 inEnabled) {
    LBAD_LOCK(d);
    if (!d) return kLBAudioDetectiveArgumentInvalid;
    d->timing = inEnabled != 0;
    d->ev_used = 0;
    return noErr;
}

OSStatus LBAudioDetectiveGetStageTimes(LBAudioDetectiveRef d, Float32* outStage1Ms, Float32* outStage2Ms,
                                       UInt32* outLaunches) {
    LBAD_LOCK(d);
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
    LBAD_LOCK(d);
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

// The sparse form of stage 2 alone, on compact frames (what the pruned stage 1 writes when more than half of the bands are
// structurally empty): per frame 128 rows of the bands that can be non-zero (LBAudioDetectiveGetCompactBands: which, in
// the order they are stored; at most 17, hence LBAD_COMPACT_FRAME_FLOATS as an upper bound).  ArgumentInvalid when the
// configuration has no such layout.  For tests and fuzzers: the batch entry points choose the layout themselves.
OSStatus LBAudioDetectiveGetCompactLayout(LBAudioDetectiveRef d, UInt32* outLeftBand, UInt32* outLiveColumns) {
    LBAD_LOCK(d);
    if (!d) return kLBAudioDetectiveArgumentInvalid;
    OSStatus st = ensure_plan(d);
    if (st != noErr) return st;
    if (!d->plan.sparse.ok || !lbad::haar_select32_supported(d->plan)) return kLBAudioDetectiveArgumentInvalid;
    if (outLeftBand) *outLeftBand = d->plan.sparse.left;
    if (outLiveColumns) *outLiveColumns = d->plan.sparse.n_cols;
    return noErr;
}

OSStatus LBAudioDetectiveGetCompactBands(LBAudioDetectiveRef d, UInt32* outBands, UInt32* outCount) {
    LBAD_LOCK(d);
    if (!d || !outCount) return kLBAudioDetectiveArgumentInvalid;
    OSStatus st = ensure_plan(d);
    if (st != noErr) return st;
    if (!d->plan.sparse.ok || !lbad::haar_select32_supported(d->plan)) return kLBAudioDetectiveArgumentInvalid;
    *outCount = d->plan.sparse.n_stored;
    if (outBands)
        for (uint32_t i = 0; i < d->plan.sparse.n_stored; ++i) outBands[i] = d->plan.sparse.stored[i];
    return noErr;
}

OSStatus LBAudioDetectiveCompactFramesToSubfingerprintsDevice(LBAudioDetectiveRef d, const Float32* inFrames, UInt64 inNumberOfFrames,
                                                              void* outPacked, Float32* outFramesHaar, void* inStream) {
    LBAD_LOCK(d);
    if (!d || (!inFrames && inNumberOfFrames) || (!outPacked && inNumberOfFrames)) return kLBAudioDetectiveArgumentInvalid;
    OSStatus st = ensure_plan(d);
    if (st != noErr) return st;
    const lbad::Plan& p = d->plan;
    if (!p.sparse.ok || !lbad::haar_select32_supported(p)) return kLBAudioDetectiveArgumentInvalid;
    hipStream_t stream = static_cast<hipStream_t>(inStream);
    if (outFramesHaar) LBAD_HIP(hipMemsetAsync(outFramesHaar, 0, inNumberOfFrames * lbad::kRowsPerFrame * p.bands * sizeof(float), stream));
    LBAD_HIP(lbad::launch_haar_select32(p, inFrames, inNumberOfFrames, static_cast<uint32_t*>(outPacked), outFramesHaar, stream, true));
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
    LBAD_GUARD_BEGIN
    return LBAudioDetectiveFingerprintClipsFormat(d, inClips, 0, inNumberOfClips, inSamplesPerClip, outBooleans);
    LBAD_GUARD_END
}

OSStatus LBAudioDetectiveFingerprintClipsFormat(LBAudioDetectiveRef d, const void* inClips, UInt32 inSampleFormat,
                                                UInt64 inNumberOfClips, UInt64 inSamplesPerClip,
                                                Boolean* outBooleans) {
    LBAD_GUARD_BEGIN
    LBAD_LOCK(d);
    if (!d || !inClips || !outBooleans || inSampleFormat > 2) return kLBAudioDetectiveArgumentInvalid;
    return lbad::fingerprint_clips_host(d, inClips, inSampleFormat, inNumberOfClips, inSamplesPerClip, outBooleans);
    LBAD_GUARD_END
}

OSStatus LBAudioDetectiveProcessPCM(LBAudioDetectiveRef d, const Float32* inSamples, UInt64 inNumberOfSamples,
                                    LBAudioDetectiveFingerprintRef* outFingerprint) {
    LBAD_GUARD_BEGIN
    LBAD_LOCK(d);
    if (!d || !outFingerprint || (!inSamples && inNumberOfSamples)) return kLBAudioDetectiveArgumentInvalid;
    OSStatus st = ensure_plan(d);
    if (st != noErr) return st;
    const uint64_t per = lbad::subfingerprint_count(inNumberOfSamples, d->window, d->stride);
    std::vector<Boolean> bools((size_t)per * d->subfp_len);
    if (per) {
        st = lbad::fingerprint_clips_host(d, inSamples, 0, 1, inNumberOfSamples, bools.data());
        if (st != noErr) return st;
    }
    *outFingerprint = lbad::fingerprint_from_bools(d, bools.data(), per);
    return noErr;
    LBAD_GUARD_END
}

OSStatus LBAudioDetectiveComparePCM(LBAudioDetectiveRef d, const Float32* inSamples1, UInt64 inCount1,
                                    const Float32* inSamples2, UInt64 inCount2, UInt32 inComparisonRange,
                                    Float32* outMatch) {  // :442-464 on PCM
    LBAD_GUARD_BEGIN
    LBAD_LOCK(d);
    if (inComparisonRange == 0) inComparisonRange = d->subfp_len;  // :443-445
    LBAudioDetectiveFingerprintRef fp1 = NULL, fp2 = NULL;
    OSStatus st = LBAudioDetectiveProcessPCM(d, inSamples1, inCount1, &fp1);
    st = LBAudioDetectiveProcessPCM(d, inSamples2, inCount2, &fp2);  // only the second status gates the compare (:453-458)
    if (st == noErr && fp1 && fp2)
        *outMatch = LBAudioDetectiveFingerprintCompareToFingerprint(fp1, fp2, inComparisonRange);
    LBAudioDetectiveFingerprintDispose(fp1);
    LBAudioDetectiveFingerprintDispose(fp2);
    return st;
    LBAD_GUARD_END
}

static OSStatus read_url(const char* path, std::vector<float>& mono, double& rate) {
    const lbad::AudioFileStatus fs = lbad::read_audio_file(path, mono, rate);
    if (fs == lbad::AudioFileStatus::NotFound) return -43;  // fnfErr, what ExtAudioFileOpenURL reports
    if (fs != lbad::AudioFileStatus::Ok) return kLBAudioDetectiveUnsupportedFile;
    return noErr;
}

OSStatus LBAudioDetectiveSetFileHopMode(LBAudioDetectiveRef d, UInt32 inMode) {
    LBAD_LOCK(d);
    if (!d || inMode > 1) return kLBAudioDetectiveArgumentInvalid;
    d->hop_mode = inMode;
    return noErr;
}

OSStatus LBAudioDetectiveSetFileTailMode(LBAudioDetectiveRef d, UInt32 inMode) {
    LBAD_LOCK(d);
    if (!d || inMode > 2) return kLBAudioDetectiveArgumentInvalid;
    d->tail_mode = inMode;
    return noErr;
}

OSStatus LBAudioDetectiveSetResamplerMode(LBAudioDetectiveRef d, UInt32 inMode) {
    LBAD_LOCK(d);
    if (!d || inMode > 2) return kLBAudioDetectiveArgumentInvalid;
    d->resampler = inMode;
    return noErr;
}

OSStatus LBAudioDetectiveReadAudioURLWithResampler(const char* inFileURL, Float64 inSampleRate,
                                                   UInt32 inResamplerMode, Float32** outSamples, UInt64* outCount,
                                                   Float64* outSampleRate) {
    LBAD_GUARD_BEGIN
    if (!inFileURL || !outSamples || !outCount || inResamplerMode > 2) return kLBAudioDetectiveArgumentInvalid;
    std::vector<float> mono, conv;
    double rate = 0.0;
    OSStatus st = read_url(inFileURL, mono, rate);
    if (st != noErr) return st;
    const std::vector<float>* src = &mono;
    if (inSampleRate > 0.0 && std::fabs(inSampleRate - rate) > 1e-9 * rate) {
        if (!lbad::resample(mono, rate, inSampleRate, inResamplerMode, conv)) return kLBAudioDetectiveArgumentInvalid;
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
    LBAD_GUARD_END
}

OSStatus LBAudioDetectiveReadAudioURL(const char* inFileURL, Float64 inSampleRate, Float32** outSamples,
                                      UInt64* outCount, Float64* outSampleRate) {
    return LBAudioDetectiveReadAudioURLWithResampler(inFileURL, inSampleRate, 0, outSamples, outCount, outSampleRate);
}

void LBAudioDetectiveFreeSamples(Float32* inSamples) { std::free(inSamples); }

// Upstream's loop over a file already converted to the processing rate (`client`), with upstream's
// bookkeeping: the window count from the length in FILE frames (:236,250-255), window i starting i * hop
// client samples in (:287-288), and the chosen treatment of the windows that reach past the end.
OSStatus LBAudioDetectiveProcessFileStream(LBAudioDetectiveRef d, const Float32* inClientSamples, UInt64 inClientCount,
                                           UInt64 inFileFrames, UInt32 inHop, LBAudioDetectiveFingerprintRef* outFingerprint) {
    LBAD_GUARD_BEGIN
    LBAD_LOCK(d);
    if (!d || !outFingerprint || (!inClientSamples && inClientCount) || inHop == 0) return kLBAudioDetectiveArgumentInvalid;
    OSStatus st = ensure_plan(d);
    if (st != noErr) return st;
    const uint32_t W = d->window;
    uint64_t frames = 0;
    if (d->stride != 0 && inFileFrames >= W) frames = ((inFileFrames - W) / d->stride) / lbad::kRowsPerFrame;   // :250,255
    if (frames == 0) {
        *outFingerprint = LBAudioDetectiveFingerprintNew(0);
        return noErr;
    }
    const uint64_t rows = frames * lbad::kRowsPerFrame;
    if (rows > (UINT64_MAX - W) / inHop) return kLBAudioDetectiveArgumentInvalid;   // rows * hop + W must not wrap
    const uint64_t need = rows * inHop + W;                // (need - W) / hop / 128 == frames
    std::vector<float> padded;
    const float* pcm = inClientSamples;
    if (need > inClientCount) {                             // stage 1 transforms every window; the tail is redone below
        padded.assign(need, 0.0f);
        std::memcpy(padded.data(), inClientSamples, sizeof(float) * inClientCount);
        pcm = padded.data();
    }
    lbad::FileTail tail;
    tail.mode = d->tail_mode;
    tail.n_client = inClientCount;
    tail.first_short = inClientCount >= W ? (inClientCount - W) / inHop + 1 : 0;
    std::vector<uint32_t> tbl;
    if (tail.mode == 2 && tail.first_short < rows) {        // nRead shrinks monotonically (:252,275: in/out argument)
        const uint32_t bands = d->bands;
        const uint64_t n_tail = rows - tail.first_short;
        tbl.resize((size_t)n_tail * (1 + 2 * bands));
        uint32_t n_read = W;
        for (uint64_t t = 0; t < n_tail; ++t) {
            const uint64_t start = (tail.first_short + t) * inHop;
            const uint64_t avail = inClientCount > start ? inClientCount - start : 0;
            if (avail < n_read) n_read = (uint32_t)avail;
            uint32_t* e = tbl.data() + (size_t)t * (1 + 2 * bands);
            e[0] = n_read;
            lbad::make_band_bounds(d->format.mSampleRate, W, n_read, d->plan.table, e + 1, e + 1 + bands);
        }
    }
    std::vector<Boolean> bools((size_t)frames * d->subfp_len);
    {
        // the plan is keyed by the hop between windows; the public stride comes back however this block is left
        struct Restore {
            LBAudioDetective* det;
            uint32_t saved;
            ~Restore() { det->stride = saved; }
        } restore{d, d->stride};
        d->stride = inHop;
        st = ensure_plan(d);
        if (st == noErr)
            st = lbad::fingerprint_clips_host(d, pcm, 0, 1, need, bools.data(), tail.mode ? &tail : nullptr, tbl.empty() ? nullptr : &tbl);
    }
    if (st != noErr) return st;
    *outFingerprint = lbad::fingerprint_from_bools(d, bools.data(), frames);
    return noErr;
    LBAD_GUARD_END
}

OSStatus LBAudioDetectiveProcessAudioURL(LBAudioDetectiveRef d, const char* inFileURL,
                                         LBAudioDetectiveFingerprintRef* outFingerprint) {  // :208-308
    LBAD_GUARD_BEGIN
    if (!d || !inFileURL || !outFingerprint) return kLBAudioDetectiveArgumentInvalid;  // :211-214
    // the pinned block, the converter buffers, the io stream and the stride are ONE set per detective: this call is
    // serialised like every other entry point (round-3 advice: the batch and pair calls were, this one was not)
    LBAD_LOCK(d);
    // ExtAudioFile decodes and converts to the client format (:229); here both happen on the device and the
    // converted samples stay there for the window loop (api_files.cpp: a batch of one file)
    return lbad::process_audio_files(d, &inFileURL, 1, outFingerprint, nullptr);
    LBAD_GUARD_END
}

// the path-taking names an Objective-C host's inline wrappers call (include/lbaudiodetective.h)
OSStatus LBAudioDetectiveProcessAudioPath(LBAudioDetectiveRef d, const char* inFilePath, LBAudioDetectiveFingerprintRef* outFingerprint) {
    return LBAudioDetectiveProcessAudioURL(d, inFilePath, outFingerprint);
}

// The file front end alone, on the device, for parity checks: the samples the window loop of ProcessAudioURL sees
// (decode + conversion to the processing rate, k_decode.hip / k_resample.hip), copied back to the host.
OSStatus LBAudioDetectiveConvertAudioURL(LBAudioDetectiveRef d, const char* inFileURL, Float32** outSamples, UInt64* outCount,
                                         UInt64* outFileFrames, Float64* outFileSampleRate) {
    LBAD_GUARD_BEGIN
    LBAD_LOCK(d);
    if (!d || !inFileURL || !outSamples || !outCount) return kLBAudioDetectiveArgumentInvalid;
    lbad::AudioPayload file;
    const lbad::AudioFileStatus fs = lbad::parse_audio_file(inFileURL, file);
    if (fs == lbad::AudioFileStatus::NotFound) return -43;
    if (fs != lbad::AudioFileStatus::Ok) return kLBAudioDetectiveUnsupportedFile;
    if (!(d->format.mSampleRate > 0.0)) return kLBAudioDetectiveArgumentInvalid;
    if (!lbad::device_ready()) return kLBAudioDetectiveDeviceUnavailable;
    std::vector<float> mono;
    OSStatus st = lbad::convert_file_on_device(d, file, d->format.mSampleRate, d->resampler, mono);
    if (st != noErr) return st;
    Float32* buf = static_cast<Float32*>(std::malloc(sizeof(Float32) * (mono.size() ? mono.size() : 1)));
    if (!buf) return kLBAudioDetectiveMemFull;
    std::memcpy(buf, mono.data(), sizeof(Float32) * mono.size());
    *outSamples = buf;
    *outCount = mono.size();
    if (outFileFrames) *outFileFrames = file.count;
    if (outFileSampleRate) *outFileSampleRate = file.sample_rate;
    return noErr;
    LBAD_GUARD_END
}

OSStatus LBAudioDetectiveCompareAudioURLs(LBAudioDetectiveRef d, const char* inFileURL1, const char* inFileURL2,
                                          UInt32 inComparisonRange, Float32* outMatch) {  // :442-464
    LBAD_GUARD_BEGIN
    LBAD_LOCK(d);
    if (!d) return kLBAudioDetectiveArgumentInvalid;
    if (inComparisonRange == 0) inComparisonRange = d->subfp_len;
    // both files in one launch chain; like upstream the status is the SECOND file's (:449-456) and outMatch is
    // written only if that one is noErr
    const char* paths[2] = {inFileURL1, inFileURL2};
    LBAudioDetectiveFingerprintRef fp[2] = {NULL, NULL};
    OSStatus sts[2] = {noErr, noErr};
    OSStatus st = lbad::process_audio_files(d, paths, 2, fp, sts);
    if (st == noErr) st = sts[1];
    if (st == noErr && fp[0] && fp[1] && outMatch)
        *outMatch = LBAudioDetectiveFingerprintCompareToFingerprint(fp[0], fp[1], inComparisonRange);
    LBAudioDetectiveFingerprintDispose(fp[0]);
    LBAudioDetectiveFingerprintDispose(fp[1]);
    return st;
    LBAD_GUARD_END
}

OSStatus LBAudioDetectiveCompareAudioPaths(LBAudioDetectiveRef d, const char* inFilePath1, const char* inFilePath2,
                                           UInt32 inComparisonRange, Float32* outMatch) {
    return LBAudioDetectiveCompareAudioURLs(d, inFilePath1, inFilePath2, inComparisonRange, outMatch);
}

// ---- streaming: chunked PCM in, the partial frame is carried across calls -----------------------------
struct LBAudioDetectiveStream {
    LBAudioDetectiveRef detective;
    std::vector<Float32> pending;          // starts at a frame boundary of the stream
    LBAudioDetectiveFingerprintRef fingerprint;
};

LBAudioDetectiveStreamRef LBAudioDetectiveStreamNew(LBAudioDetectiveRef inDetective) {
    if (!inDetective) return NULL;
    LBAudioDetectiveStream* s = new (std::nothrow) LBAudioDetectiveStream();
    if (!s) return NULL;
    s->detective = inDetective;
    s->fingerprint = LBAudioDetectiveFingerprintNew(0);
    if (!s->fingerprint) { delete s; return NULL; }
    return s;
}

void LBAudioDetectiveStreamDispose(LBAudioDetectiveStreamRef inStream) {
    if (!inStream) return;
    LBAudioDetectiveFingerprintDispose(inStream->fingerprint);
    delete inStream;
}

OSStatus LBAudioDetectiveStreamPush(LBAudioDetectiveStreamRef s, const Float32* inSamples, UInt64 inNumberOfSamples,
                                    UInt32* outNewSubfingerprints) {
    LBAD_GUARD_BEGIN
    if (outNewSubfingerprints) *outNewSubfingerprints = 0;
    if (!s || (!inSamples && inNumberOfSamples)) return kLBAudioDetectiveArgumentInvalid;
    LBAudioDetective* d = s->detective;
    LBAD_LOCK(d);
    OSStatus st = ensure_plan(d);
    if (st != noErr) return st;
    s->pending.insert(s->pending.end(), inSamples, inSamples + inNumberOfSamples);
    // same rule as the whole-buffer path (:250-255): frame f exists once (L - W) / stride >= 128 (f + 1)
    const uint64_t ready = lbad::subfingerprint_count(s->pending.size(), d->window, d->stride);
    if (ready == 0) return noErr;
    const uint64_t hop = (uint64_t)lbad::kRowsPerFrame * d->stride;
    const uint64_t use = (uint64_t)d->window + ready * hop;            // yields exactly `ready` frames
    std::vector<Boolean> bools((size_t)ready * d->subfp_len);
    st = lbad::fingerprint_clips_host(d, s->pending.data(), 0, 1, use, bools.data());
    if (st != noErr) return st;
    for (uint64_t f = 0; f < ready; ++f) {
        UInt32 len = d->subfp_len;
        LBAudioDetectiveFingerprintSetSubfingerprintLength(s->fingerprint, &len);
        LBAudioDetectiveFingerprintAddSubfingerprint(s->fingerprint, bools.data() + (size_t)f * d->subfp_len);
    }
    s->pending.erase(s->pending.begin(), s->pending.begin() + (size_t)(ready * hop));
    if (outNewSubfingerprints) *outNewSubfingerprints = (UInt32)ready;
    return noErr;
    LBAD_GUARD_END
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

OSStatus LBAudioDetectiveSynthRaggedCorpusDevice(UInt32 inSeed, UInt64 inFirstEntry, UInt64 inNumberOfEntries,
                                                 const UInt32* inOffsets, UInt64 inTotalSubfingerprints,
                                                 UInt32 inSubfingerprintLength, void* outPacked, void* inStream) {
    if (!outPacked || !inOffsets || inSubfingerprintLength == 0 || inSubfingerprintLength > LBAD_MAX_SUBFINGERPRINT_LENGTH)
        return kLBAudioDetectiveArgumentInvalid;
    if (!lbad::device_ready()) return kLBAudioDetectiveDeviceUnavailable;
    LBAD_HIP(lbad::launch_synth_ragged(inSeed, inFirstEntry, inNumberOfEntries, inOffsets, inTotalSubfingerprints,
                                       inSubfingerprintLength, static_cast<uint32_t*>(outPacked),
                                       static_cast<hipStream_t>(inStream)));
    return noErr;
}

SInt32 LBAudioDetectiveDeviceCount(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
OSStatus LBAudioDetectiveProbeShaderClock(void* inStream, UInt32 inMicroseconds, Float64* outMegahertz) {
    if (!outMegahertz || inMicroseconds == 0 || inMicroseconds > 1000000u) return kLBAudioDetectiveArgumentInvalid;
    if (!lbad::device_ready()) return kLBAudioDetectiveDeviceUnavailable;
    hipStream_t stream = static_cast<hipStream_t>(inStream);
    unsigned long long* d = nullptr;
    unsigned long long h[2] = {0, 0};
    LBAD_HIP(hipMalloc(reinterpret_cast<void**>(&d), sizeof(h)));
    OSStatus st = lbad::hip_status(lbad::launch_clock_probe(inMicroseconds, d, stream), "clock probe", __LINE__);
    if (st == noErr) st = lbad::hip_status(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, stream), "copy", __LINE__);
    if (st == noErr) st = lbad::hip_status(hipStreamSynchronize(stream), "sync", __LINE__);
    (void)hipFree(d);
    if (st != noErr) return st;
    *outMegahertz = h[1] ? 100.0 * (double)h[0] / (double)h[1] : 0.0;
    return noErr;
}

OSStatus LBAudioDetectiveDeviceSet(SInt32 inDevice) {
    LBAD_HIP(hipSetDevice(inDevice));
    return noErr;
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
