// api_files.cpp -- the file entry points as one GPU workload.
//
// Upstream fingerprints a file by seeking and reading it window by window through ExtAudioFile
// (LBAudioDetective/LBAudioDetective.m:208-308); its test suite does that 200 times per test
// (LBAudioDetectiveTests/LBAudioDetectiveTests.m:57-91).  Here any number of files goes through ONE launch chain:
//   host    files read into one pinned block and parsed in place by a pool of reader threads, each uploading its part
//   device  payload decode (k_decode.hip) and sample-rate conversion (k_resample.hip) of every file straight into
//           its slot of ONE float32 clip; stage 1 over that clip; the files' end-of-file rows; stage 2; 32 bytes per
//           sub-fingerprint come back
// The converted samples never leave the device.  Slots: with a hop of h samples a frame of 128 windows advances
// G = 128 h samples; file f gets frames_f + ceil(W / G) whole frames of the clip, so that its windows lie on the
// clip's window grid and the windows that straddle two files fall into frames nobody reads.  Files with
// different hops (different file rates in hop mode 1) form separate clips.
#include "internal.hpp"
#include "audiofile.hpp"

#include <fcntl.h>
#include <pthread.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cmath>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#ifdef LBAD_EXP_FILE_TIMES   // experiment builds: where a batch call spends its host time (stderr, microseconds)
#include <chrono>
#define LBAD_T(name) const auto name = std::chrono::steady_clock::now()
#define LBAD_US(a, b) (long)std::chrono::duration_cast<std::chrono::microseconds>((b) - (a)).count()
#else
#define LBAD_T(name)
#endif

namespace lbad {
namespace {

struct Job {
    AudioPayload a;
    ResamplePlan rp;
    OSStatus st = noErr;
    uint32_t hop = 0;
    uint64_t frames = 0;       // sub-fingerprints of the file
    uint64_t n_client = 0;     // samples at the processing rate
    uint64_t slot_frames = 0, frame0 = 0;   // frames of the clip the file owns, first of them
    uint64_t bytes0 = 0, dec0 = 0;          // offsets of its payload / decoded samples in the batch buffers
    uint64_t file_off = 0, file_size = 0;   // the whole file inside the pinned block
    uint64_t first_short = 0;
    size_t tbl0 = 0, tbl_n = 0;             // tail mode 2: its table inside the batch's table block (words)
};

// restore the detective's public stride however the call ends (the plan is keyed by the hop between windows)
struct StrideGuard {
    LBAudioDetective* d;
    uint32_t saved;
    explicit StrideGuard(LBAudioDetective* det) : d(det), saved(det->stride) {}
    ~StrideGuard() { d->stride = saved; }
};

OSStatus grow_pinned(void** ptr, size_t* cap, size_t bytes) {
    if (*cap >= bytes) return noErr;
    if (*ptr) (void)hipHostFree(*ptr);
    *ptr = nullptr;
    *cap = 0;
    const size_t want = bytes + bytes / 4;
    LBAD_HIP(hipHostMalloc(ptr, want, hipHostMallocDefault));
    *cap = want;
    return noErr;
}

// The threads that read and parse the files of a batch: started once (a std::thread costs ~20 us to create, sixteen of them
// were a third of the read phase), parked on a condition variable between calls, shared by every detective of the process.
class ReadPool {
public:
    // never destroyed: its threads sleep on a condition variable until the process ends.  In a child of fork() the
    // threads do not exist; the child runs its tasks on the calling thread.
    static ReadPool& get() {
        static ReadPool* pool = [] {
            (void)pthread_atfork(nullptr, nullptr, [] { forked().store(true); });
            return new ReadPool;
        }();
        return *pool;
    }
    // fn(0) .. fn(n_tasks - 1), each exactly once, on the pool's threads and the caller's; returns when all are done.
    // A task that throws (std::bad_alloc from a vector inside the caller's lambda) does not take the process down and
    // does not leave workers running on a dead frame: the first exception is kept, every remaining task is skipped,
    // ALL threads are awaited, and the exception is rethrown here, on the calling thread (whose LBAD_GUARD turns it
    // into a status).
    void run(size_t n_tasks, const std::function<void(size_t, bool)>& fn) {
        if (forked().load()) {
            for (size_t i = 0; i < n_tasks; ++i) fn(i, false);
            return;
        }
        std::lock_guard<std::mutex> one_batch(batch_);
        {
            std::lock_guard<std::mutex> g(m_);
            fn_ = &fn;
            n_ = n_tasks;
            next_.store(0);
            pending_ = threads_.size();
            failed_.store(false);
            error_ = nullptr;
            ++generation_;
        }
        start_.notify_all();
        for (size_t i; (i = next_.fetch_add(1)) < n_tasks;) run_one(fn, i, false);
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [&] { return pending_ == 0; });
        fn_ = nullptr;
        if (error_) {
            std::exception_ptr e = error_;
            error_ = nullptr;
            g.unlock();
            std::rethrow_exception(e);
        }
    }
    size_t threads() const { return threads_.size() + 1; }

private:
    ReadPool() {
        unsigned n = std::thread::hardware_concurrency();
        n = n > 16 ? 16 : n;
        try {
            for (unsigned i = 1; i < n; ++i) threads_.emplace_back([this] { work(); });
        } catch (const std::system_error&) {   // fewer threads than wanted: the caller's thread takes the rest
        }
    }
    static std::atomic<bool>& forked() {
        static std::atomic<bool> f{false};
        return f;
    }
    void work() {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(size_t, bool)>* fn;
            size_t n;
            {
                std::unique_lock<std::mutex> g(m_);
                start_.wait(g, [&] { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
                fn = fn_;
                n = n_;
            }
            for (size_t i; (i = next_.fetch_add(1)) < n;) run_one(*fn, i, true);
            {
                std::lock_guard<std::mutex> g(m_);
                --pending_;
            }
            done_.notify_one();
        }
    }
    void run_one(const std::function<void(size_t, bool)>& fn, size_t i, bool on_worker) {
        if (failed_.load(std::memory_order_relaxed)) return;       // a task failed: the rest of the batch is void anyway
        try {
            fn(i, on_worker);
        } catch (...) {
            std::lock_guard<std::mutex> g(m_);
            if (!error_) error_ = std::current_exception();
            failed_.store(true);
        }
    }
    std::atomic<bool> failed_{false};
    std::exception_ptr error_;
    std::vector<std::thread> threads_;
    std::mutex m_, batch_;
    std::condition_variable start_, done_;
    const std::function<void(size_t, bool)>* fn_ = nullptr;
    size_t n_ = 0, pending_ = 0;
    std::atomic<size_t> next_{0};
    uint64_t generation_ = 0;
    bool stop_ = false;
};

#ifndef LBAD_MIN_RUN_MB
#define LBAD_MIN_RUN_MB 16
#endif
constexpr uint64_t kMinRunBytes = (uint64_t)LBAD_MIN_RUN_MB << 20;
#ifndef LBAD_RUN_RAMP
#define LBAD_RUN_RAMP 1
#endif
constexpr uint64_t kFirstRunBytes = 4ull << 20;     // the shortest first run of a pipelined call
constexpr size_t kAlign = 256;
size_t align_up(size_t v) { return (v + kAlign - 1) & ~(kAlign - 1); }

// A group of files on its way through the device: what must stay alive until its results have been unpacked.
struct Pending {
    bool active = false;
    std::vector<size_t> idx;            // the files (indices into jobs)
    std::vector<uint32_t> tbl;          // tail mode 2 tables (source of an asynchronous copy)
    std::vector<FileDesc> descs;        // descriptors (likewise)
    uint32_t* packed = nullptr;         // pinned: where the packed sub-fingerprints land
    int slot = 0;
};

OSStatus finish_group(LBAudioDetective* d, std::vector<Job>& jobs, Pending& p, LBAudioDetectiveFingerprintRef* out);

// one clip: jobs[idx[0..n)] share `hop`.  Everything is enqueued on the detective's stream -- decode, conversion, the window
// loop's two kernels, the copy of the packed results into the slot's pinned block -- and `p` keeps what the copies read;
// finish_group() waits for the stream and builds the fingerprints.
OSStatus enqueue_group(LBAudioDetective* d, std::vector<Job>& jobs, std::vector<size_t>&& idx_in, uint32_t hop, int slot, Pending& p) {
    p = Pending();
    p.idx = std::move(idx_in);
    p.slot = slot;
    const std::vector<size_t>& idx = p.idx;
    const uint32_t W = d->window, bands = d->bands;
    const uint64_t G = (uint64_t)kRowsPerFrame * hop;
    const uint64_t gap = (W + G - 1) / G;
    uint64_t total_frames = 0;
    size_t dec_total = 0, tbl_words = 0;
    const bool stale = d->hop_mode == 1 && d->tail_mode == 2;
    for (size_t i : idx) {
        Job& j = jobs[i];
        j.frame0 = total_frames;
        j.slot_frames = j.frames + gap;
        total_frames += j.slot_frames;
        j.dec0 = dec_total;
        dec_total += (size_t)align_up(j.a.total_frames * sizeof(float)) / sizeof(float);
        j.first_short = j.n_client >= W ? (j.n_client - W) / hop + 1 : 0;
        const uint64_t rows = j.frames * kRowsPerFrame;
        if (stale && j.first_short < rows) {
            j.tbl0 = tbl_words;
            j.tbl_n = (size_t)(rows - j.first_short) * (1 + 2 * bands);
            tbl_words += j.tbl_n;
        }
    }
    if (total_frames > 0xFFFFFFFFull / kRowsPerFrame) return kLBAudioDetectiveArgumentInvalid;
    const uint64_t T = total_frames * G + W;                       // samples of the clip: exactly total_frames frames
    const size_t packed_bytes = (size_t)total_frames * LBAD_PACKED_BYTES;

    OSStatus st = grow_device(&d->d_rs_in, &d->d_rs_in_cap, dec_total * sizeof(float));
    if (st == noErr) st = grow_device(&d->d_rs_out, &d->d_rs_out_cap, T * sizeof(float));
    if (st == noErr) st = grow_device(reinterpret_cast<void**>(&d->d_io_packed), &d->d_io_packed_cap, packed_bytes);
    if (st == noErr && tbl_words) st = grow_device(&d->d_rs_tail, &d->d_rs_tail_cap, tbl_words * sizeof(uint32_t));
    void** h_packed = slot ? &d->h_packed_b : &d->h_packed;
    size_t* h_packed_cap = slot ? &d->h_packed_b_cap : &d->h_packed_cap;
    if (st == noErr) st = grow_pinned(h_packed, h_packed_cap, packed_bytes);
    if (st != noErr) return st;
    LBAD_T(g0);
    hipStream_t stream = d->io_stream;                             // the payload bytes are on their way on this stream
    float* pcm = static_cast<float*>(d->d_rs_out);
    LBAD_HIP(hipMemsetAsync(pcm, 0, T * sizeof(float), stream));   // the slots' zero padding

    // tail mode 2: nRead shrinks monotonically over the short windows (:252,275: in/out argument)
    std::vector<uint32_t>& tbl = p.tbl;
    tbl.assign(tbl_words, 0u);
    for (size_t i : idx) {
        const Job& j = jobs[i];
        if (!j.tbl_n) continue;
        const uint64_t n_tail = j.frames * kRowsPerFrame - j.first_short;
        uint32_t n_read = W;
        for (uint64_t t = 0; t < n_tail; ++t) {
            const uint64_t start = (j.first_short + t) * hop;
            const uint64_t avail = j.n_client > start ? j.n_client - start : 0;
            if (avail < n_read) n_read = (uint32_t)avail;
            uint32_t* e = tbl.data() + j.tbl0 + (size_t)t * (1 + 2 * bands);
            e[0] = n_read;
            make_band_bounds(d->format.mSampleRate, W, n_read, d->plan.table, e + 1, e + 1 + bands);
        }
    }
    if (tbl_words) LBAD_HIP(hipMemcpyAsync(d->d_rs_tail, tbl.data(), tbl_words * sizeof(uint32_t), hipMemcpyHostToDevice, stream));

    // one descriptor per file for the table-driven kernels: decode, conversion and (tail mode 1) the short rows of
    // ALL files are one launch each
    std::vector<FileDesc>& descs = p.descs;
    descs.assign(idx.size(), FileDesc());
    std::vector<FileTail> tails;
    uint64_t max_units = 0, max_out = 0, max_short = 0;
    const uint32_t mode = d->resampler;
    bool any_sinc = false;
    for (size_t k = 0; k < idx.size(); ++k) {
        const Job& j = jobs[idx[k]];
        FileDesc& f = descs[k];
        f.kind = (uint32_t)j.a.kind; f.channels = j.a.channels; f.bits = j.a.bits;
        f.flags = (j.a.is_float ? 1u : 0u) | (j.a.little ? 2u : 0u);
        f.bytes_off = j.bytes0; f.total_frames = j.a.total_frames;
        f.dec_off = j.dec0; f.first = j.a.first; f.n_in = j.a.count;
        f.out_off = j.frame0 * G;
        const uint64_t slot_len = j.slot_frames * G;                 // what lies beyond is the next file's
        f.n_write = j.n_client < slot_len ? j.n_client : slot_len;
        f.mode = j.rp.mode; f.copy = j.rp.copy ? 1u : 0u;
        f.ratio = j.rp.ratio; f.scale = j.rp.scale; f.half = j.rp.half;
        st = device_phase(d, j.rp.copy ? nullptr : j.rp.phases, stream, f);
        if (st != noErr) return st;
        f.row_begin = j.frame0 * kRowsPerFrame; f.rows = j.frames * kRowsPerFrame; f.first_short = j.first_short;
        const uint64_t units = j.a.kind == AudioPayload::Ima4 ? j.a.total_frames / 64 : j.a.total_frames;
        if (units > max_units) max_units = units;
        if (f.n_write > max_out) max_out = f.n_write;
        if (f.first_short < f.rows && f.rows - f.first_short > max_short) max_short = f.rows - f.first_short;
        any_sinc = any_sinc || (!j.rp.copy && mode < 2);
        if (d->hop_mode == 1 && d->tail_mode == 2) {
            FileTail t;
            t.mode = 2;
            t.first_short = j.first_short;
            t.n_client = j.n_client;
            t.d_tbl = j.tbl_n ? static_cast<const uint32_t*>(d->d_rs_tail) + j.tbl0 : nullptr;
            t.row_begin = f.row_begin;
            t.rows = f.rows;
            t.pcm_begin = f.out_off;
            tails.push_back(t);
        }
    }
    const size_t desc_bytes = descs.size() * sizeof(FileDesc);
    st = grow_device(&d->d_rs_desc, &d->d_rs_desc_cap, desc_bytes);
    if (st != noErr) return st;
    const FileDesc* d_files = static_cast<const FileDesc*>(d->d_rs_desc);
    LBAD_HIP(hipMemcpyAsync(d->d_rs_desc, descs.data(), desc_bytes, hipMemcpyHostToDevice, stream));
    const double* d_table = nullptr;
    uint64_t table_n = 0;
    int table_res = 0;
    if (any_sinc) {
        const ResamplePlan* rp = nullptr;
        for (size_t i : idx)
            if (!jobs[i].rp.copy) { rp = &jobs[i].rp; break; }
        table_n = rp->table->size();
        table_res = rp->table_res;
        if (!d->d_rs_table[mode]) {
            LBAD_HIP(hipMalloc(reinterpret_cast<void**>(&d->d_rs_table[mode]), table_n * sizeof(double)));
            LBAD_HIP(hipMemcpyAsync(d->d_rs_table[mode], rp->table->data(), table_n * sizeof(double), hipMemcpyHostToDevice, stream));
        }
        d_table = d->d_rs_table[mode];
    }
    LBAD_HIP(launch_decode_batch(d_files, (uint32_t)descs.size(), max_units,
                                 static_cast<const uint8_t*>(slot ? d->d_rs_bytes_b : d->d_rs_bytes), static_cast<float*>(d->d_rs_in), stream));
    if (d->bytes_free[slot]) LBAD_HIP(hipEventRecord(d->bytes_free[slot], stream));   // the slot's payloads may be replaced from here on
    LBAD_HIP(launch_resample_batch(d_files, (uint32_t)descs.size(), max_out, static_cast<const float*>(d->d_rs_in), table_res,
                                   d_table, table_n, pcm, stream));
    if (d->hop_mode == 1 && d->tail_mode == 1 && max_short) {
        FileTail t;
        t.mode = 3;
        t.d_files = d_files;
        t.n_files = (uint32_t)descs.size();
        t.max_rows = max_short;
        tails.push_back(t);
    }

    {
        StrideGuard guard(d);
        d->stride = hop;
        st = ensure_plan(d);
        if (st == noErr)
            st = fingerprint_clips_device(d, pcm, 0, 1, T, d->d_io_packed, nullptr, nullptr, stream,
                                          tails.empty() ? nullptr : tails.data(), tails.size());
    }
    if (st != noErr) return st;
    p.packed = static_cast<uint32_t*>(*h_packed);
    LBAD_HIP(hipMemcpyAsync(p.packed, d->d_io_packed, packed_bytes, hipMemcpyDeviceToHost, stream));
    if (!d->packed_done[slot]) LBAD_HIP(hipEventCreateWithFlags(&d->packed_done[slot], hipEventDisableTiming));
    LBAD_HIP(hipEventRecord(d->packed_done[slot], stream));
    p.active = true;
#ifdef LBAD_EXP_FILE_TIMES
    LBAD_T(g1);
    fprintf(stderr, "  group of %zu files: tables + launches %ld us\n", idx.size(), LBAD_US(g0, g1));
#endif
    return noErr;
}

// the group's device work is awaited -- ITS event, not the stream: the next run's kernels, enqueued behind it, keep the device
// busy while this group's packed results become upstream's Boolean rows
OSStatus finish_group(LBAudioDetective* d, std::vector<Job>& jobs, Pending& p, LBAudioDetectiveFingerprintRef* out) {
    if (!p.active) return noErr;
    p.active = false;
    const std::vector<size_t>& idx = p.idx;
    const uint32_t* packed = p.packed;
    LBAD_T(g1);
    LBAD_HIP(hipEventSynchronize(d->packed_done[p.slot]));
    LBAD_T(g2);
    // 32 bytes per sub-fingerprint back into upstream's Boolean rows: a few files per task on the reader pool
    auto unpack = [&](size_t k_begin, size_t k_end) {
        std::vector<Boolean> bools;
        for (size_t k = k_begin; k < k_end; ++k) {
            const size_t i = idx[k];
            const Job& j = jobs[i];
            bools.assign((size_t)j.frames * d->subfp_len, 0);
            for (uint64_t s = 0; s < j.frames; ++s)
                LBAudioDetectiveUnpackSubfingerprint(packed + (j.frame0 + s) * LBAD_PACKED_WORDS, d->subfp_len,
                                                     bools.data() + (size_t)s * d->subfp_len);
            out[i] = fingerprint_from_bools(d, bools.data(), j.frames);
        }
    };
    constexpr size_t kPerTask = 4;
    if (idx.size() < 4 * kPerTask) {
        unpack(0, idx.size());
    } else {
        ReadPool::get().run((idx.size() + kPerTask - 1) / kPerTask, [&](size_t task, bool) {
            unpack(task * kPerTask, (task + 1) * kPerTask < idx.size() ? (task + 1) * kPerTask : idx.size());
        });
    }
#ifdef LBAD_EXP_FILE_TIMES
    LBAD_T(g3);
    fprintf(stderr, "  group of %zu files: wait for the device %ld us, unpack %ld us\n", idx.size(), LBAD_US(g1, g2), LBAD_US(g2, g3));
#endif
    return noErr;
}

// the files [run_b, n) of one pinned block, already on their way to the device: framing, then one clip per hop
void fail_group(std::vector<Job>& jobs, const std::vector<size_t>& idx, OSStatus st, LBAudioDetectiveFingerprintRef* out) {
    for (size_t k : idx) {
        jobs[k].st = st;
        if (out[k]) { LBAudioDetectiveFingerprintDispose(out[k]); out[k] = NULL; }
    }
}

// (the run's LAST group is left on the device -- `pending` -- for the caller to finish after it has read the next run)
void process_run(LBAudioDetective* d, std::vector<Job>& jobs, size_t run_b, size_t n, uint64_t budget, std::vector<bool>& done,
                 LBAudioDetectiveFingerprintRef* out, int slot, Pending& pending) {
    // framing per file (:236,250-255), then one clip per hop value, cut where the inter-stage buffer would overflow
    const double rate = d->format.mSampleRate;
    OSStatus st = noErr;
    for (size_t i = run_b; i < n; ++i) {
        Job& j = jobs[i];
        if (j.st != noErr) { done[i] = true; continue; }
        if (d->hop_mode == 0) {
            j.hop = d->stride;
            j.frames = subfingerprint_count(j.n_client, d->window, d->stride);
        } else {
            // what upstream does: the length (:236) and the seek offsets (:287-288) are in FILE frames while
            // each read asks for windowSize CLIENT frames, so the hop is analysisStride file frames =
            // analysisStride * rate / file_rate client samples and the window count comes from the file length
            uint32_t hop = (uint32_t)std::llround((double)d->stride * rate / j.a.sample_rate);
            j.hop = hop < 1 ? 1 : hop;
            j.frames = j.a.count >= d->window ? ((j.a.count - d->window) / d->stride) / kRowsPerFrame : 0;
        }
        if (j.frames == 0) {
            out[i] = LBAudioDetectiveFingerprintNew(0);
            done[i] = true;
        }
    }
    for (size_t i = run_b; i < n; ++i) {
        if (done[i]) continue;
        const uint32_t hop = jobs[i].hop;
        const uint64_t gap = (d->window + (uint64_t)kRowsPerFrame * hop - 1) / ((uint64_t)kRowsPerFrame * hop);
        std::vector<size_t> idx;
        uint64_t frames = 0;
        for (size_t k = i; k < n; ++k) {
            if (done[k] || jobs[k].hop != hop) continue;
            if (!idx.empty() && (frames + jobs[k].frames + gap > budget || idx.size() >= 65535)) break;
            idx.push_back(k);
            frames += jobs[k].frames + gap;
            done[k] = true;
        }
        // the groups of a run share the slot's pinned result block: an earlier group is finished before the next is enqueued
        if (pending.active) {
            st = finish_group(d, jobs, pending, out);
            if (st != noErr) fail_group(jobs, pending.idx, st, out);
        }
        std::vector<size_t> members = idx;
        st = enqueue_group(d, jobs, std::move(idx), hop, slot, pending);
        if (st != noErr) {
            (void)hipStreamSynchronize(d->io_stream);          // nothing may still read what `pending` owns
            pending.active = false;
            fail_group(jobs, members, st, out);
        }
    }
}

}  // namespace

OSStatus process_audio_files(LBAudioDetective* d, const char* const* paths, size_t n, LBAudioDetectiveFingerprintRef* out,
                             OSStatus* statuses) {
    if (!d || !paths || !out) return kLBAudioDetectiveArgumentInvalid;
    for (size_t i = 0; i < n; ++i) out[i] = NULL;
    const double rate = d->format.mSampleRate;
    if (!(rate > 0.0)) return kLBAudioDetectiveArgumentInvalid;
    OSStatus st = noErr;
    LBAD_T(t0);
    std::vector<Job> jobs(n);

    // sizes first: a missing or unreadable file is reported as such whether or not a device exists (like
    // ExtAudioFileOpenURL); everything after that needs the GPU
    auto sizes = [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            Job& j = jobs[i];
            if (!paths[i]) { j.st = kLBAudioDetectiveArgumentInvalid; continue; }       // :211-214
            struct stat sb;
            if (::stat(paths[i], &sb) != 0) { j.st = -43; continue; }                     // fnfErr
            const long long sz = (long long)sb.st_size;
            if (sz <= 0) { j.st = kLBAudioDetectiveUnsupportedFile; continue; }
            j.file_size = (uint64_t)sz;
        }
    };
    constexpr size_t kStatsPerTask = 128;
    if (n < 4 * kStatsPerTask) {
        sizes(0, n);
    } else {
        ReadPool::get().run((n + kStatsPerTask - 1) / kStatsPerTask, [&](size_t task, bool) {
            sizes(task * kStatsPerTask, (task + 1) * kStatsPerTask < n ? (task + 1) * kStatsPerTask : n);
        });
    }
    bool any = false;
    for (size_t i = 0; i < n; ++i) any = any || (jobs[i].st == noErr && jobs[i].file_size);
    if (any) {
        st = ensure_plan(d);
        if (st == noErr && !d->io_stream) st = hip_status(hipStreamCreateWithFlags(&d->io_stream, hipStreamNonBlocking), "stream", __LINE__);
        if (st != noErr) {
            // no usable device: a file that could never be read is still reported as such, the others get the
            // device's status
            for (size_t i = 0; i < n; ++i) {
                Job& j = jobs[i];
                if (j.st != noErr) continue;
                AudioPayload probe;
                j.st = parse_audio_file(paths[i], probe) == AudioFileStatus::Ok ? st : kLBAudioDetectiveUnsupportedFile;
            }
            any = false;
        }
    }
    // the files of a call go through in runs of at most 512 MB of file bytes: read straight into ONE pinned block
    // (pooled threads), parsed in place, uploaded part by part
    LBAD_T(t1);
    // ... and two runs are in flight: while the device works on run i the pool reads run i + 1 into the OTHER pinned block,
    // and run i's results are unpacked while the device works on run i + 1.  A call of many files is cut into at least
    // eight runs (of at least 16 MB) for that; upstream's own workload is 200 files per test (LBAudioDetectiveTests.m:57-91).
    uint64_t all_bytes = 0;
    for (size_t i = 0; i < n; ++i)
        if (jobs[i].st == noErr) all_bytes += align_up(jobs[i].file_size);
    uint64_t kRunBytes = 512ull << 20;
    // Round 6: the FIRST runs are short.  Nothing runs on the device until the first run has been read and uploaded -- with
    // eight equal runs the device sat idle for the first 8 ms of a 57 ms call (profiles/r06_file_pipeline.txt: 3.6 ms of
    // cold reads, then 161 MB over PCIe).  The first run is a sixty-fourth of the call (at least kFirstRunBytes), every
    // run twice its predecessor until the eighth is reached: the device starts after about 2 ms and each run's kernels
    // cover the reading of the next, larger one.
    uint64_t ramp_bytes = 0;
    if (d->file_pipeline) {
        const uint64_t eighth = all_bytes / 8;
        kRunBytes = eighth < kMinRunBytes ? kMinRunBytes : (eighth < kRunBytes ? eighth : kRunBytes);
        if (LBAD_RUN_RAMP && all_bytes / 64 >= kFirstRunBytes / 2) {
            ramp_bytes = all_bytes / 64 < kFirstRunBytes ? kFirstRunBytes : all_bytes / 64;
            if (ramp_bytes >= kRunBytes) ramp_bytes = 0;
        }
    }
    Pending in_flight;                  // the previous run's last group
    int slot = 0;
    struct Drain {                      // however the call ends, nothing enqueued may outlive what it reads
        LBAudioDetective* d;
        ~Drain() {
            if (d->up_stream) (void)hipStreamSynchronize(d->up_stream);
            if (d->io_stream) (void)hipStreamSynchronize(d->io_stream);
        }
    } drain{d};
    std::vector<bool> done(n, false);
    const uint64_t frame_bytes = (uint64_t)kRowsPerFrame * d->bands * sizeof(float);
    const uint64_t budget = d->scratch_limit / frame_bytes ? d->scratch_limit / frame_bytes : 1;
    for (size_t run_b = 0; any && run_b < n;) {
        size_t run_e = run_b;
        uint64_t total = 0;
        const uint64_t run_limit = ramp_bytes ? ramp_bytes : kRunBytes;
        if (ramp_bytes) ramp_bytes = 2 * ramp_bytes >= kRunBytes ? 0 : 2 * ramp_bytes;
        while (run_e < n && (run_e == run_b || total + align_up(jobs[run_e].file_size) <= run_limit)) {
            if (jobs[run_e].st == noErr) {
                jobs[run_e].file_off = total;
                total += align_up(jobs[run_e].file_size);
            }
            ++run_e;
        }
        void** h_files = slot ? &d->h_files_b : &d->h_files;
        size_t* h_files_cap = slot ? &d->h_files_b_cap : &d->h_files_cap;
        // (sized for the call's largest run at once: the short first runs must not make the later ones re-allocate)
        const uint64_t largest = all_bytes < kRunBytes ? all_bytes : kRunBytes;
        const uint64_t block = d->file_pipeline && total < largest ? largest : total;
        st = total ? grow_pinned(h_files, h_files_cap, block) : noErr;
        void** d_bytes = slot ? &d->d_rs_bytes_b : &d->d_rs_bytes;
        size_t* d_bytes_cap = slot ? &d->d_rs_bytes_b_cap : &d->d_rs_bytes_cap;
        if (st == noErr && total) st = grow_device(d_bytes, d_bytes_cap, block);
        // the run's payloads go up on their own stream, beside the previous run's kernels: they wait for the decode kernel
        // that read this slot two runs ago, and this run's kernels wait for them
        if (st == noErr && !d->up_stream) st = hip_status(hipStreamCreateWithFlags(&d->up_stream, hipStreamNonBlocking), "stream", __LINE__);
        for (int k = 0; k < 2 && st == noErr; ++k) {
            if (!d->up_done[k]) st = hip_status(hipEventCreateWithFlags(&d->up_done[k], hipEventDisableTiming), "event", __LINE__);
            if (st == noErr && !d->bytes_free[k]) {
                st = hip_status(hipEventCreateWithFlags(&d->bytes_free[k], hipEventDisableTiming), "event", __LINE__);
                if (st == noErr) st = hip_status(hipEventRecord(d->bytes_free[k], d->io_stream), "event", __LINE__);
            }
        }
        if (st == noErr) st = hip_status(hipStreamWaitEvent(d->up_stream, d->bytes_free[slot], 0), "event", __LINE__);
        if (st != noErr) {
            for (size_t i = run_b; i < run_e; ++i)
                if (jobs[i].st == noErr) jobs[i].st = st;
            run_b = run_e;
            continue;
        }
        uint8_t* stage = static_cast<uint8_t*>(*h_files);
        int device = 0;
        (void)hipGetDevice(&device);
        std::atomic<int> upload_error{0};
        // every worker reads its files (read(2) straight into the pinned block), parses them in place and sends ITS part of
        // the block on its way: the uploads of the first workers overlap the reads of the last
        auto read_range = [&](size_t b, size_t e, bool own_thread) {
            if (own_thread) (void)hipSetDevice(device);
            uint64_t lo = ~0ull, hi = 0;
            for (size_t i = b; i < e; ++i) {
                Job& j = jobs[i];
                if (j.st != noErr) continue;
                if (j.file_off < lo) lo = j.file_off;
                if (j.file_off + j.file_size > hi) hi = j.file_off + j.file_size;
                const int fd = ::open(paths[i], O_RDONLY | O_CLOEXEC);
                uint64_t got = 0;
                while (fd >= 0 && got < j.file_size) {
                    const ssize_t r = ::read(fd, stage + j.file_off + got, j.file_size - got);
                    if (r <= 0) break;
                    got += (uint64_t)r;
                }
                if (fd >= 0) ::close(fd);
                if (got != j.file_size) { j.st = fd >= 0 ? kLBAudioDetectiveUnsupportedFile : -43; continue; }
                const AudioFileStatus fs = parse_audio_bytes(stage + j.file_off, j.file_size, j.a);
                if (fs != AudioFileStatus::Ok) { j.st = kLBAudioDetectiveUnsupportedFile; continue; }
                if (!resample_plan(j.a.count, j.a.sample_rate, rate, d->resampler, j.rp)) { j.st = kLBAudioDetectiveArgumentInvalid; continue; }
                j.n_client = j.a.count == 0 ? 0 : j.rp.n_out;
                j.bytes0 = j.file_off + j.a.off;
            }
            if (hi > lo) {
                const hipError_t err = hipMemcpyAsync(static_cast<uint8_t*>(*d_bytes) + lo, stage + lo, hi - lo, hipMemcpyHostToDevice, d->up_stream);
                if (err != hipSuccess) upload_error.store((int)err);
            }
        };
        const size_t count = run_e - run_b;
        if (count < 4) {
            read_range(run_b, run_e, false);
        } else {
            ReadPool& pool = ReadPool::get();
            const size_t per = (count + 2 * pool.threads() - 1) / (2 * pool.threads());    // about two tasks per thread
            const size_t n_tasks = (count + per - 1) / per;
            pool.run(n_tasks, [&](size_t task, bool own_thread) {
                const size_t b = run_b + task * per, e = b + per < run_e ? b + per : run_e;
                read_range(b, e, own_thread);
            });
        }
        LBAD_T(t2);
        if (upload_error.load() == 0) {
            hipError_t e = hipEventRecord(d->up_done[slot], d->up_stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(d->io_stream, d->up_done[slot], 0);
            if (e != hipSuccess) upload_error.store((int)e);
        }
        if (upload_error.load() != 0) {
            st = hip_status((hipError_t)upload_error.load(), "payload upload", __LINE__);
            (void)hipStreamSynchronize(d->up_stream);
            (void)hipStreamSynchronize(d->io_stream);
            for (size_t i = run_b; i < run_e; ++i)
                if (jobs[i].st == noErr) jobs[i].st = st;
            run_b = run_e;
            continue;
        }
        LBAD_T(t3);
        Pending mine;
        process_run(d, jobs, run_b, run_e, budget, done, out, slot, mine);
        // the previous run has had the time this one took to read: collect it now, behind this run's launches
        if (in_flight.active) {
            const OSStatus fst = finish_group(d, jobs, in_flight, out);
            if (fst != noErr) fail_group(jobs, in_flight.idx, fst, out);
        }
        in_flight = std::move(mine);
        if (d->file_pipeline) slot ^= 1;
        else if (in_flight.active) {
            const OSStatus fst = finish_group(d, jobs, in_flight, out);
            if (fst != noErr) fail_group(jobs, in_flight.idx, fst, out);
        }
#ifdef LBAD_EXP_FILE_TIMES
        LBAD_T(t4);
        fprintf(stderr, "run of %zu files, %llu bytes: sizes + plan %ld us, read + parse %ld us, upload call %ld us, groups %ld us\n",
                run_e - run_b, (unsigned long long)total, LBAD_US(t0, t1), LBAD_US(t1, t2), LBAD_US(t2, t3), LBAD_US(t3, t4));
#endif
        run_b = run_e;
    }
    if (in_flight.active) {
        const OSStatus fst = finish_group(d, jobs, in_flight, out);
        if (fst != noErr) fail_group(jobs, in_flight.idx, fst, out);
    }
    OSStatus first = noErr;
    for (size_t i = 0; i < n; ++i) {
        if (statuses) statuses[i] = jobs[i].st;
        if (first == noErr && jobs[i].st != noErr) first = jobs[i].st;
    }
    return statuses ? noErr : first;
}

}  // namespace lbad

extern "C" {

OSStatus LBAudioDetectiveProcessAudioURLs(LBAudioDetectiveRef inDetective, const char* const* inFileURLs,
                                          UInt32 inCount, LBAudioDetectiveFingerprintRef* outFingerprints,
                                          OSStatus* outStatuses) {
    LBAD_GUARD_BEGIN
    LBAD_LOCK(inDetective);
    if (!inDetective || (!inFileURLs && inCount) || (!outFingerprints && inCount)) return kLBAudioDetectiveArgumentInvalid;
    if (inCount == 0) return noErr;
    return lbad::process_audio_files(inDetective, inFileURLs, inCount, outFingerprints, outStatuses);
    LBAD_GUARD_END
}

OSStatus LBAudioDetectiveSetFilePipeline(LBAudioDetectiveRef inDetective, UInt32 inEnabled) {
    LBAD_LOCK(inDetective);
    if (!inDetective) return kLBAudioDetectiveArgumentInvalid;
    inDetective->file_pipeline = inEnabled != 0;
    return noErr;
}

}  // extern "C"
