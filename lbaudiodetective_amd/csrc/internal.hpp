// internal.hpp -- shared declarations of the HIP library (not installed; the ABI is include/lbaudiodetective.h)
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <new>
#include <stdexcept>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/lbaudiodetective.h"

namespace lbad {

constexpr uint32_t kRowsPerFrame = 128;  // LBAudioDetective.m:25
constexpr uint32_t kPackedWords = LBAD_PACKED_WORDS;
constexpr uint32_t kMaxBands = 64;
constexpr uint32_t kSparseFrameDwMax = 128 * 17;       // the largest compact frame (Plan::Sparse): 128 rows of at most 16 + 1 stored bands
constexpr uint32_t kMinWindow = 16;
constexpr uint32_t kMaxWindow = 8192;

OSStatus hip_status(hipError_t e, const char* what, int line);
#define LBAD_HIP(expr)                                                   \
    do {                                                                 \
        OSStatus st__ = ::lbad::hip_status((expr), #expr, __LINE__);     \
        if (st__ != noErr) return st__;                                  \
    } while (0)

// Host allocations sized by caller or file data (decoded audio, resampler output, zero padding) can fail:
// nothing may unwind through the C boundary.  memFullErr is MacErrors.h's -108.
#define LBAD_GUARD_BEGIN try {
#define LBAD_GUARD_END                                                    \
    }                                                                     \
    catch (const std::bad_alloc&) { return kLBAudioDetectiveMemFull; }    \
    catch (const std::exception&) { return kLBAudioDetectiveArgumentInvalid; }

bool device_ready();
// Per-device facts and one-time set-up, keyed by the CURRENT device (a process may use several).
constexpr int kMaxDevices = 64;
int current_device();                       // hipGetDevice, -1 on failure
int device_cu_count();                      // CUs of the current device (256 when the query fails)
// One slot per device: returns true when `value` differs from what was recorded for the current device
// (and records it) -- "has this function's attribute / table been set up on this device for this size?"
struct PerDevice {
    size_t seen[kMaxDevices] = {};
    bool changed(size_t value) {
        const int d = current_device();
        if (d < 0 || d >= kMaxDevices) return true;
        if (seen[d] == value) return false;
        seen[d] = value;
        return true;
    }
};

// ---- per-configuration device plan ---------------------------------------------------------
struct BandTable {
    std::vector<uint32_t> indices;  // bands + 1
    std::vector<uint32_t> lo, hi;   // bin bounds per band
    uint32_t kmin = 0, kmax = 0;    // union of [lo, hi) over non-empty bands (kmin == kmax: nothing read)
    // Where a window's power terms lie in LDS (k_rows_full.hip, k_rows_stream2.hip): band b's terms in bin order from word
    // term_at[b], the bands one after the other with a gap in front of a band where its first term would share a bank
    // (word mod 32) with an earlier band's -- the lanes that add the bands' terms side by side then never meet on a bank.
    // term_end: first word behind the last band.  ordered == false (a table whose bands overlap or are not in bin order;
    // make_band_table never makes one): terms by bin number, term_at[b] = lo[b] - kmin.
    std::vector<uint32_t> term_at;
    uint32_t term_end = 0;
    bool ordered = true;
    // 2048-sample windows on k_rows_full.hip: bit (q * 16 + u) * 2 + half is set when NO lane's bin of that place in the split
    // pass (k_rows_full.hip: slot_work) is read by a band -- the kernel has an instance that leaves the default table's out
    uint64_t unread_terms16 = 0;
};

// host-side, double precision; mirrors LBAudioDetective.m:361-371,382-383
void make_band_table(double sample_rate, uint32_t window, uint32_t bands, BandTable& out);
// bin bounds of every band when ComputeFrequencies is handed n_frames != window (a short read, :281,382-383);
// reads stay inside the window-sized buffer
void make_band_bounds(double sample_rate, uint32_t window, uint32_t n_frames, const BandTable& table, uint32_t* lo,
                      uint32_t* hi);
// master twiddle table exp(-2 pi i k / W), k in [0, W/2)
void make_twiddles(uint32_t W, std::vector<float>& re, std::vector<float>& im);

struct Plan {
    double sample_rate = 0;
    uint32_t window = 0, stride = 0, bands = 0, subfp_len = 0;
    uint32_t log2w = 0;
    uint32_t keep = 0;  // wavelets whose sign pair survives the truncation to subfp_len Booleans
    BandTable table;
    // device copies
    float* d_tw = nullptr;        // [W/2] re then [W/2] im
    uint32_t* d_bands = nullptr;  // [bands] lo, [bands] hi, [bands] divisor as float bits, ... (api_detective.cpp: eight rows + 1 word)
    float* d_bin_const = nullptr; // per-bin twiddles of the pruned kernel (only when pruned_ok)
    bool pruned_ok = false;
    bool full_ok = false;         // k_rows_full.hip applies
    bool stream_ok = false;       // k_rows_stream.hip applies (also uses d_claim)
    bool stream2_ok = false;      // k_rows_stream2.hip applies (preferred over k_rows_full.hip when the clip length is even)
    uint32_t* d_claim = nullptr;  // its per-XCD claim counters (8 words)
    // Structurally empty bands (round 4): a band whose bin range is empty is +0.0 in every window (17 of the 32
    // bands at 44.1 kHz / 1024).  `sparse.ok`: 32 bands and at most ONE live band among the left sixteen -- stage 1 then
    // writes compact frames (128 rows of the `n_stored` bands that can be non-zero: the live ones of the right sixteen in
    // ascending order, then the left half's one live band)
    // and stage 2 runs its sparse form (k_haar_select32.hip): one thread per row, and only the columns of the row
    // transform that can be non-zero go through the column transform and the select.
    struct Sparse {
        bool ok = false;
        uint32_t left = 32;            // the live band of the left half (32: none)
        uint32_t n_cols = 0;           // columns of the row transform's output that can be non-zero ...
        uint8_t cols[32] = {};         // ... in ascending ordered position
        uint32_t n_stored = 0;         // bands a compact frame's row holds
        uint8_t stored[17] = {};       // ... which ones, by position in the row
        uint8_t pos_right[16];         // position of band 16 + j in the row (0xFF: structurally empty, not stored)
        uint8_t pos_left = 0xFF;       // position of the left half's live band
        uint32_t frame_dw() const { return 128u * n_stored; }
    } sparse;
    // measurement knobs of the generic stage-1 kernel (LBAudioDetectiveSetKernelTuning): waves per workgroup
    // (0 = automatic) and whether the per-lane twiddle cache is used
    uint32_t tune_waves = 0;
    bool tune_cache = true;
    bool valid = false;
};

// End-of-file treatment of upstream's file loop for one file inside a float32 clip: the file's windows start at
// row `row_begin` / sample `pcm_begin` of the clip; windows from `first_short` on (counted from the file's first
// window) belong to reads that cannot be met in full.
struct FileDesc;
struct FileTail {
    uint32_t mode = 0;          // 1: nothing read -> all-zero rows; 2: partial reads over the stale spectrum; 3: mode 1 for a
                                //    whole batch in one launch (d_files / n_files / max_rows below, one FileTail in all)
    const FileDesc* d_files = nullptr;
    uint32_t n_files = 0;
    uint64_t max_rows = 0;
    uint64_t first_short = 0;   // first such window of the file
    uint64_t n_client = 0;      // samples the file really has at the processing rate
    const uint32_t* d_tbl = nullptr;   // mode 2: per window [n_read, lo[bands], hi[bands]] on the device
    uint64_t row_begin = 0;     // first row of the file inside the clip
    uint64_t rows = 0;          // rows of the file (0: all rows of the clip)
    uint64_t pcm_begin = 0;     // first sample of the file inside the clip
};

// ---- kernel launchers (k_*.hip) -------------------------------------------------------------
// windows -> frame rows.  frames: [n_clips * frames_per_clip][128][bands]
// fmt: 0 float32, 1 int16, 2 int32 samples
hipError_t launch_fft_bands(const Plan& plan, const void* d_pcm, uint32_t fmt, uint64_t n_clips,
                            uint64_t samples_per_clip, uint32_t frames_per_clip, float* d_frames, hipStream_t stream);
// frame rows -> packed sub-fingerprints.  d_haar (optional) receives the decomposed frames.
hipError_t launch_haar_select(const Plan& plan, float* d_frames, uint64_t n_frames, uint32_t* d_packed,
                              float* d_haar_out, hipStream_t stream);
// specialised stage 1 (k_rows_pruned.hip): 1024-sample windows whose bands read only bins 0..21
bool rows_pruned_supported(const Plan& plan);
void rows_pruned_constants(std::vector<float>& out);
// compact: rows of 16 floats (plan.sparse), else rows of 32
hipError_t launch_rows_pruned(const Plan& plan, const float* d_bin_const, const void* d_pcm, uint32_t fmt,
                              uint64_t n_clips, uint64_t samples_per_clip, uint32_t frames_per_clip, float* d_frames,
                              hipStream_t stream, bool compact = false);

// specialised stage 1 without pruning (k_rows_full.hip): 1024- and 2048-sample windows, any band table
bool rows_full_supported(const Plan& plan);
bool rows_full_supported_fmt(const Plan& plan, uint32_t fmt);   // strides other than 64: float32 input only
hipError_t launch_rows_full(const Plan& plan, const void* d_pcm, uint32_t fmt, uint64_t n_clips,
                            uint64_t samples_per_clip, uint32_t frames_per_clip, float* d_frames, hipStream_t stream);

// specialised stage 1 for 4096-sample windows (k_rows_stream.hip): a wave walks the windows of a frame and keeps
// the sub-transforms consecutive windows share.  Needs an even samples_per_clip (aligned sample pairs).
bool rows_stream_supported(const Plan& plan);
hipError_t launch_rows_stream(const Plan& plan, const void* d_pcm, uint32_t fmt, uint64_t n_clips,
                              uint64_t samples_per_clip, uint32_t frames_per_clip, float* d_frames, hipStream_t stream);

// the same plan for 2048-sample windows (k_rows_stream2.hip; the reference's default configuration)
bool rows_stream2_supported(const Plan& plan);
hipError_t launch_rows_stream2(const Plan& plan, const void* d_pcm, uint32_t fmt, uint64_t n_clips,
                               uint64_t samples_per_clip, uint32_t frames_per_clip, float* d_frames, hipStream_t stream);

// specialised stage 2 (k_haar_select32.hip): 128 x 32 frames, keep <= 128
bool haar_select32_supported(const Plan& plan);
// compact: d_frames holds plan.sparse's rows of 16 floats (the sparse form; d_haar_out must be zeroed by the caller)
hipError_t launch_haar_select32(const Plan& plan, const float* d_frames, uint64_t n_frames, uint32_t* d_packed,
                                float* d_haar_out, hipStream_t stream, bool compact = false);
void plan_sparse(Plan& plan);    // fills plan.sparse from plan.table

// end-of-file chain of upstream's file loop, tail mode "stale" (k_file_tail.hip); d_tbl: per window
// [n_read, lo[bands], hi[bands]]
hipError_t launch_empty_rows(const Plan& p, float* d_rows, uint64_t n_rows, hipStream_t stream);
hipError_t launch_file_tail(const Plan& plan, const float* d_pcm, uint64_t n_client, uint32_t hop, uint64_t first_short,
                            uint32_t n_tail, const uint32_t* d_tbl, float* d_frames, hipStream_t stream);

// generic matrix ops behind the Frame API
// one file of a batch for the table-driven kernels (k_decode.hip, k_resample.hip, k_file_tail.hip)
struct FileDesc {
    uint32_t kind, channels, bits, flags;      // AudioPayload::Kind; flags: 1 float samples, 2 little endian
    uint64_t bytes_off, total_frames;          // payload inside the batch's byte block; frames it decodes to
    uint64_t dec_off, first, n_in;             // decoded samples inside the batch's block; first / count that are valid
    uint64_t out_off, n_write;                 // the file's slot in the clip; samples to write there
    uint32_t mode, copy;                       // converter model; 1: rates equal, copy
    double ratio, scale, half;                 // ResamplePlan
    uint64_t row_begin, rows, first_short;     // the file's rows in the clip; first window whose read is short
    // rational position (audiofile.hpp, PhaseTable): q != 0, the phases' weights on the device
    uint64_t ph_p, ph_q;
    int32_t ph_m_min;
    uint32_t ph_m_span;
    const int32_t* ph_first;
    const uint32_t* ph_count;
    const double* ph_wsum;
    const double* ph_w;
};
struct PhaseTable;
// the device copy of a phase table, made on first use and kept by the detective
struct DevPhase {
    const PhaseTable* host = nullptr;
    int32_t* first = nullptr;
    uint32_t* count = nullptr;
    double* wsum = nullptr;
    double* w = nullptr;
};
OSStatus device_phase(struct ::LBAudioDetective* d, const PhaseTable* host, hipStream_t stream, FileDesc& f);
hipError_t launch_decode_batch(const FileDesc* d_files, uint32_t n_files, uint64_t max_units, const uint8_t* d_bytes,
                               float* d_decoded, hipStream_t stream);
hipError_t launch_resample_batch(const FileDesc* d_files, uint32_t n_files, uint64_t max_out, const float* d_decoded, int res,
                                 const double* d_table, uint64_t table_n, float* d_pcm, hipStream_t stream);
hipError_t launch_empty_rows_batch(const Plan& p, const FileDesc* d_files, uint32_t n_files, uint64_t max_rows, float* d_frames,
                                   hipStream_t stream);
hipError_t launch_haar2d_generic(float* d_m, float* d_tmp, uint32_t rows, uint32_t cols, hipStream_t stream);
hipError_t launch_extract_generic(const float* d_m, uint32_t n, uint32_t n_wavelets, uint8_t* d_out,
                                  hipStream_t stream);

// compare
// slot layout: entries[e][s][8 words]; writes per-entry score (optional) and atomically maxes the key
hipError_t launch_compare_slots(const uint32_t* d_entries, uint64_t n_entries, uint32_t n_sub, uint32_t subfp_len,
                                const uint32_t* d_query, uint32_t n_query, uint32_t range, uint64_t index_base,
                                float* d_scores, unsigned long long* d_key, hipStream_t stream);
// one fingerprint against one (slot layout, a = the side with at least as many sub-fingerprints): the score's
// float bits are atomicMax-ed into *d_out_bits (zeroed by the caller)
hipError_t launch_compare_pair(const uint32_t* d_a, uint32_t n1, const uint32_t* d_b, uint32_t n2, uint32_t subfp_len,
                               uint32_t range, unsigned int* d_out_bits, hipStream_t stream);
// plane layout (tight bitstream, 16-byte planes); supported shapes only
bool planes_supported(uint32_t subfp_len, uint32_t n_sub);
uint32_t planes_per_entry(uint32_t subfp_len, uint32_t n_sub);
hipError_t launch_pack_planes(const uint32_t* d_slots, uint64_t n_entries, uint32_t n_sub, uint32_t subfp_len,
                              uint4* d_planes, uint64_t plane_stride, uint64_t first, hipStream_t stream);
hipError_t launch_compare_planes_generic(const uint4* d_planes, uint64_t plane_stride, uint64_t n_entries,
                                         uint32_t n_sub, uint32_t subfp_len, const uint32_t* d_query,
                                         uint32_t n_query, uint32_t range, uint64_t index_base, float* d_scores,
                                         unsigned long long* d_key, hipStream_t stream);
// specialised scan (200-Boolean sub-fingerprints, query and entries of equal count <= 8)
bool planes_fast_supported(uint32_t subfp_len, uint32_t n_sub, uint32_t n_query);
uint32_t planes_fast_const_words(uint32_t n_sub);
void build_plane_query(const uint32_t* q_slots, uint32_t n_sub, uint32_t range, std::vector<uint32_t>& out);
// d_ticket / host_out_dev / seq: optional tail for the host-synchronous query: d_key is then an array of
// kScanSlots per-workgroup slots, the last workgroup reduces them and hands the key to pinned host memory;
// pass nullptr to get the plain device-key behaviour
constexpr uint32_t kScanSlots = 512;
hipError_t launch_compare_planes_fast(const uint4* d_planes, uint64_t plane_stride, uint64_t n_entries,
                                      uint32_t n_sub, const uint32_t* d_qc, uint64_t index_base, float* d_scores,
                                      unsigned long long* d_key, hipStream_t stream, unsigned int* d_ticket = nullptr,
                                      unsigned long long* host_out_dev = nullptr, unsigned long long seq = 0);
uint32_t plane_query_words();
hipError_t launch_compare_planes_batch(const uint4* d_planes, uint64_t plane_stride, uint64_t n_entries, uint32_t n_sub,
                                       const uint32_t* d_qblocks, uint32_t n_queries, uint64_t index_base,
                                       unsigned long long* d_keys, hipStream_t stream);
void pack_fingerprint(const struct ::LBAudioDetectiveFingerprint* fp, std::vector<uint32_t>& out);

// ragged corpus (k_sliding.hip): a stream of 32-byte sub-fingerprint records, entries of any length back to back
bool sliding_supported(uint32_t subfp_len);
uint32_t sliding_query_words(uint32_t n_query);
void build_sliding_query(const Boolean* bools, uint32_t n_query, uint32_t subfp_len, uint32_t range,
                         std::vector<uint32_t>& out);
// d_off_new: ABSOLUTE record positions of the n_new new entries (n_new + 1 values, the first one = slot 0's position)
hipError_t launch_pack_records(const uint32_t* d_slots, uint64_t n_new_pos, const uint32_t* d_off_new, uint64_t n_new,
                               uint32_t first_entry, uint4* d_recs, hipStream_t stream);
// records read from a file: derived fields (table row, place inside the entry) recomputed from d_off, reserved bits and
// pairs beyond the length cleared (old_layout: the round-3 "LBADCRP2" record)
hipError_t launch_restamp_records(uint4* d_recs, const uint32_t* d_off, uint64_t n_entries, uint64_t n, uint32_t subfp_len,
                                  bool old_layout, hipStream_t stream);
hipError_t launch_synth_ragged(uint32_t seed, uint64_t first_entry, uint64_t n_entries, const uint32_t* d_off,
                               uint64_t n_pos, uint32_t subfp_len, uint32_t* d_out, hipStream_t stream);
// shape of a scan: workgroups (one per CU) and the tasks of either kind each of them owns
struct SlideShape {
    uint32_t grid = 0, chunk_a = 0, chunk_b = 0;
};
constexpr uint32_t kSlideMaxGrid = 1024;
SlideShape sliding_shape(uint64_t tasks_a, uint64_t tasks_b, uint32_t n_q = 1);
// how many of n_left queries of one length a single launch takes (1, 2, 4; 8 for the systolic scan of short queries)
uint32_t sliding_queries_per_launch(uint32_t n_query, uint32_t ne_max, uint32_t n_left);
bool sliding_multi(uint32_t n_query, uint32_t ne_max);    // a batch of such queries takes compare_short_multi_kernel (keys max-ed in place)
// One launch of the scan: n_q queries of one length.  d_queries: their blocks (build_sliding_query without the header),
// (n_query + 1) * 16 words each, on the device; h_query: the same block of a single query on the host -- short enough it
// travels in the kernel's arguments and d_queries may be null.  d_acc (n_q words) and d_ticket are zero between scans
// (the scan leaves them so); d_keys receives the n_q results.
struct SlideScan {
    const uint32_t* d_queries = nullptr;
    const uint32_t* h_query = nullptr;
    uint32_t n_q = 1;
    unsigned long long* d_acc = nullptr;
    unsigned int* d_ticket = nullptr;
    unsigned long long* d_keys = nullptr;
    uint32_t key_pos[8] = {};     // query i's key goes to d_keys[key_pos[i]]
};
constexpr uint32_t kSlideQueryArgSubs = 47;   // longest query that travels as a kernel argument
size_t sliding_plan_words(uint64_t capacity);
// the plan of a query length (where every workgroup's run of entries starts) into d_plan
hipError_t launch_sliding_plan(const uint32_t* d_off, uint64_t n_entries, uint32_t n_query, uint32_t b_min, const SlideShape& sh,
                               uint32_t* d_plan, hipStream_t stream);
// tasks_a / tasks_b: groups of four sliding offsets over the entries longer / not longer than the query; d_query: the
// block build_sliding_query made; zero_rec: index of an all-zero record
bool sliding_short(uint32_t n_query, uint32_t ne_max);    // the systolic scan of short queries applies (no plan needed)
hipError_t launch_compare_sliding(const uint4* d_recs, uint64_t n_pos, const uint32_t* d_off, uint64_t n_entries, uint32_t ne_max,
                                  uint32_t zero_rec, uint64_t tasks_a, uint64_t tasks_b, const SlideShape& sh, const uint32_t* d_plan,
                                  uint32_t subfp_len, const SlideScan& scan, uint32_t n_query, uint32_t range,
                                  uint64_t index_base, unsigned int* d_score_bits, hipStream_t stream, bool bound_pruning = true,
                                  float prune_from = 0.7f, uint32_t b_min = 0);
// b_min > 0 (kSlideSplitBelow): the scan is SPLIT -- entries of fewer sub-fingerprints that are not longer than the query go
// through the systolic scan (a second launch over the records), everything else through the task kernel; tasks_b and the
// plan then count only the "B" entries of at least b_min sub-fingerprints
constexpr uint32_t kSlideSplitBelow = 16;
// limits of a ragged corpus (the key carries a 32-bit index, the scan's claim cursor and its record offsets want a little slack)
constexpr uint64_t kMaxRaggedEntries = 0xFFFF0000ull;
constexpr uint64_t kMaxRaggedRecords = 0xFFFFFF00ull;
constexpr uint32_t kRecordSlack = 8;   // records allocated behind a ragged corpus' capacity (zero: over-read + the zero record)

// measurement: ticks of the shader clock and of the constant 100 MHz clock over ~usec microseconds (2 words)
hipError_t launch_clock_probe(uint32_t usec, unsigned long long* d_out, hipStream_t stream);
// synthetic data
hipError_t launch_synth_clips(uint32_t seed, uint64_t first, uint64_t n_clips, uint32_t rate_hz, uint32_t n_samples,
                              uint32_t stereo, float* d_out, hipStream_t stream);
hipError_t launch_synth_corpus(uint32_t seed, uint64_t first, uint64_t n_entries, uint32_t n_sub, uint32_t subfp_len,
                               uint32_t* d_out, hipStream_t stream);

// ---- shared between the api_*.cpp files ------------------------------------------------------------------
uint64_t subfingerprint_count(uint64_t n_samples, uint32_t window, uint32_t stride);
OSStatus ensure_plan(struct ::LBAudioDetective* d);
OSStatus grow_device(void** ptr, size_t* cap, size_t bytes);
// the batch hot path on device memory; tails: end-of-file treatment of files laid out inside ONE float32 clip
OSStatus fingerprint_clips_device(struct ::LBAudioDetective* d, const void* d_pcm, uint32_t fmt, uint64_t n_clips,
                                  uint64_t samples_per_clip, uint32_t* d_packed, float* d_raw, float* d_haar,
                                  hipStream_t stream, const FileTail* tails = nullptr, size_t n_tails = 0);
::LBAudioDetectiveFingerprintRef fingerprint_from_bools(const struct ::LBAudioDetective* d, const Boolean* bools, uint64_t per);
// the file entry points (api_files.cpp): decode, conversion and the window loop of n files in one launch chain per
// hop value; statuses (optional) receives every file's status
OSStatus process_audio_files(struct ::LBAudioDetective* d, const char* const* paths, size_t n,
                             ::LBAudioDetectiveFingerprintRef* out, OSStatus* statuses);

}  // namespace lbad

// ---- handle layouts ------------------------------------------------------------------------
struct LBAudioDetectiveFingerprint {
    uint32_t length = 0;
    uint32_t count = 0;
    std::vector<Boolean> data;  // count * length
};

struct LBAudioDetectiveFrame {
    uint32_t max_rows = 0;
    uint32_t n_rows = 0;
    uint32_t row_length = 0;
    std::vector<std::vector<Float32>> rows;  // max_rows slots, ragged like the reference
};

struct LBAudioDetective {
    AudioStreamBasicDescription format;
    uint32_t subfp_len;
    uint32_t window;
    uint32_t stride;
    uint32_t bands;
    uint32_t variant = 0;
    uint32_t tune_waves = 0;  // copied into the plan
    bool tune_cache = true;
    uint32_t hop_mode = 1;   // file entry points: 0 = hop in processing-rate samples, 1 = upstream's file-frame hop
    uint32_t tail_mode = 1;  // hop mode 1, windows past the end of the file: 0 zero-filled, 1 nothing read, 2 stale spectrum
    uint32_t resampler = 0;  // 0 long Kaiser sinc, 1 short sinc, 2 linear interpolation
    lbad::Plan plan;         // lazily rebuilt when the configuration changes
    float* d_frames = nullptr;  // frame rows between stage 1 and stage 2
    uint64_t d_frames_cap = 0;  // in floats
    // bytes of HBM the inter-stage buffer may take.  512 MiB = 32 768 frames per chunk: the configs[1] pass of 500 000
    // frames runs as 16 chunks and is 2.5 % FASTER than as one (20.05 against 20.57 ms; 16 GiB was the default until
    // round 3), below 128 MiB the launches start to cost (tools/exp/scratch_chunks.py)
    uint64_t scratch_limit = 512ull << 20;
    // persistent buffers of the one-off (host in, host out) entry points; they only grow
    void* d_io_pcm = nullptr;
    size_t d_io_pcm_cap = 0;
    uint32_t* d_io_packed = nullptr;
    size_t d_io_packed_cap = 0;
    void* h_io = nullptr;        // pinned staging for small calls
    size_t h_io_cap = 0;
    hipStream_t io_stream = nullptr;
    // converter state of the file entry points: input / output samples and the two kernel tables, grown on demand
    void* d_rs_bytes = nullptr;       // the file's payload as read (file batches: slot 0)
    size_t d_rs_bytes_cap = 0;
    void* d_rs_bytes_b = nullptr;     // file batches, slot 1: run i + 1's payloads go up while run i is decoded
    size_t d_rs_bytes_b_cap = 0;
    hipStream_t up_stream = nullptr;  // the uploads of a file batch (beside io_stream, which runs the kernels)
    hipEvent_t up_done[2] = {nullptr, nullptr};      // behind a run's uploads, on up_stream
    hipEvent_t bytes_free[2] = {nullptr, nullptr};   // behind the decode kernel that read the slot's payloads, on io_stream
    hipEvent_t packed_done[2] = {nullptr, nullptr};  // behind the copy of a group's packed results into the slot's pinned block
    void* d_rs_in = nullptr;          // decoded mono samples at the file's rate
    size_t d_rs_in_cap = 0;
    void* d_rs_out = nullptr;
    size_t d_rs_out_cap = 0;
    double* d_rs_table[2] = {nullptr, nullptr};
    std::vector<lbad::DevPhase> d_phases;   // phase tables of the rational rate pairs met so far
    void* d_rs_desc = nullptr;        // per-file descriptors of a file batch
    size_t d_rs_desc_cap = 0;
    void* d_rs_tail = nullptr;        // tail-mode-2 tables of a file batch
    size_t d_rs_tail_cap = 0;
    // a file batch goes through in runs, two in flight (api_files.cpp): while the device works on run i the host reads
    // run i + 1 into the other pinned block, and run i's results are unpacked while the device works on run i + 1
    void* h_files = nullptr;          // pinned block a run of files is read into (slot 0)
    size_t h_files_cap = 0;
    void* h_packed = nullptr;         // pinned landing area of a run's packed results (slot 0)
    size_t h_packed_cap = 0;
    void* h_files_b = nullptr;        // slot 1
    size_t h_files_b_cap = 0;
    void* h_packed_b = nullptr;
    size_t h_packed_b_cap = 0;
    bool file_pipeline = true;        // LBAudioDetectiveSetFilePipeline(…, 0): one run at a time (measurement)
    // optional per-stage timing (hipEvents on the caller's stream)
    bool timing = false;
    std::vector<hipEvent_t> ev;   // 3 per chunk: start, after stage 1, after stage 2
    size_t ev_used = 0;
    // Concurrent use (round 3).  The claim counters of the plan, the inter-stage rows and the io / converter buffers
    // are ONE set per detective: `mutex` serialises the host side of every entry point, and a batch call that arrives
    // on another stream than its predecessor first waits (on the device, hipStreamWaitEvent) for `done` -- the
    // predecessor's last kernel.  Two threads with two streams therefore get correct results, one after the other;
    // for overlap use one detective per stream.  Nothing is recorded or awaited while the stream is being captured
    // into a graph (a replay is the caller's to order).
    std::recursive_mutex mutex;
    hipEvent_t done = nullptr;
    hipStream_t done_stream = nullptr;
    bool done_valid = false;
};

// first statement of every entry point that touches a detective's state
#define LBAD_LOCK(d) std::unique_lock<std::recursive_mutex> lbad_lock_; if (d) lbad_lock_ = std::unique_lock<std::recursive_mutex>((d)->mutex)

struct LBAudioDetectiveCorpus {
    uint32_t subfp_len = 0;
    uint32_t n_sub = 0;
    uint64_t capacity = 0;
    uint64_t count = 0;
    uint32_t variant = 0;
    uint4* d_planes = nullptr;    // plane layout [n_planes][capacity]
    uint32_t n_planes = 0;
    uint32_t* d_query = nullptr;  // query block on the device
    uint32_t* h_query = nullptr;  // pinned staging copy of it
    uint32_t query_cap = 0;       // in words
    unsigned long long* d_key = nullptr;
    // host-synchronous query without memset / copy / stream synchronisation (api_corpus.cpp)
    unsigned long long* d_fast_key = nullptr;    // zero between queries
    unsigned int* d_ticket = nullptr;
    unsigned long long* h_out = nullptr;         // pinned, host-coherent: [0] key, [1] sequence number
    unsigned long long* h_out_dev = nullptr;     // its device address
    unsigned long long seq = 0;
    hipStream_t stream = nullptr;
    bool appended = false;                       // entries were appended since the last polled query
    hipEvent_t append_event = nullptr;           // recorded behind the latest append on ITS stream
    // ragged form (LBAudioDetectiveCorpusNewRagged): entries of any length as a stream of 32-byte records
    bool ragged = false;
    uint4* d_recs = nullptr;                     // 2 x uint4 per record
    uint64_t rec_capacity = 0;                   // records
    uint64_t n_pos = 0;                          // records stored
    uint32_t* d_off = nullptr;                   // capacity + 1 record positions (entry e = [off[e], off[e + 1]))
    std::vector<uint32_t> h_off;                 // count + 1
    uint32_t ne_max = 0;                         // longest entry
    std::map<uint32_t, uint64_t> len_hist;       // entries per length: the scan's task totals for any query length
    // the scan's plan for ONE query length (k_sliding.hip): rebuilt when the length, the entries or the grid change
    uint32_t* d_plan = nullptr;
    uint32_t plan_nq = 0, plan_grid = 0, plan_bmin = 0;
    bool bound_pruning = true;     // top-1 scans of a ragged corpus may drop passes that cannot reach the best match so far (exact)
    float prune_from = 0.7f;       // ... once a match of at least this score is known (LBAudioDetectiveCorpusSetBoundPruningThreshold)
    uint64_t plan_count = 0;
    hipEvent_t plan_built = nullptr;             // behind the plan's kernels, on plan_stream
    hipStream_t plan_stream = nullptr;
    // key block of the sharded query (api_rccl.cpp), made with the corpus so that the collective call never allocates
    unsigned long long* d_shard_keys = nullptr;
    unsigned long long* h_shard_keys = nullptr;
    // ring of query slots in h_query / d_query (ragged scan): slot size in words, one event per slot, queries so far
    size_t query_slot_words = 0;
    hipEvent_t query_ev[8] = {};
    // per ring slot: the scan's running maxima (8 words) and its ticket, ZERO between scans -- the scan's last workgroup
    // leaves them so (k_sliding.hip: ScanOut); 16 words per slot
    unsigned long long* d_scan_out = nullptr;
    bool scan_out_dirty = false;                 // a scan's launch failed: clear the words before the next one
    std::mutex shard_lock;                       // the sharded query's key block is one per corpus (api_rccl.cpp)
    bool shard_stale = false;                    // a sharded query timed out: work may still be queued behind the key block
    hipEvent_t shard_stale_event = nullptr;      // ... recorded behind that work when the call gave up (the stream may be gone by the next call)
    uint64_t query_seq = 0;
};
