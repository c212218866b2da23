// api_corpus.cpp -- device-resident reference-fingerprint corpus and its top-1 query.
// Scales the best-match loop of LBAudioDetectiveTests/LBAudioDetectiveTests.m:57-91 (one original
// against N candidates, strict '<', first maximum wins) to an HBM-resident database.
#include "internal.hpp"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>

namespace lbad {
namespace {

// make room for `words` words in the corpus' query staging pair (device + pinned)
OSStatus reserve_query(LBAudioDetectiveCorpus* c, size_t words) {
    if (c->query_cap >= words) return noErr;
    if (c->d_query) (void)hipFree(c->d_query);
    if (c->h_query) (void)hipHostFree(c->h_query);
    c->d_query = nullptr;
    c->h_query = nullptr;
    c->query_cap = 0;
    LBAD_HIP(hipMalloc(reinterpret_cast<void**>(&c->d_query), words * sizeof(uint32_t)));
    LBAD_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_query), words * sizeof(uint32_t), hipHostMallocDefault));
    c->query_cap = (uint32_t)words;
    return noErr;
}

// ragged corpus: the sliding scan of k_sliding.hip (any query length, any entry lengths).  A launch's query blocks travel
// through a ring of kQuerySlots pinned + device slots, one event per slot: a call waits only for the scan that used
// its slot kQuerySlots launches ago (long done), not for the stream -- back-to-back queries leave no gap on the GPU.
// Round 5: a single query of up to kSlideQueryArgSubs sub-fingerprints travels in the kernel's argument segment (no copy
// node), the result words are cleared by the previous scan's last workgroup (no memset node), and up to four queries of
// ONE length share a pass over the corpus (eight in the systolic scan of short queries).
constexpr uint32_t kQuerySlots = 8;
constexpr uint32_t kScanOutWords = 16;      // per slot: 8 running maxima, the ticket, padding

// tasks of either kind for queries of nq sub-fingerprints, from the histogram of entry lengths (Fp.m:123-136: an entry
// longer than the query slides the query along itself, any other entry slides along the query)
// (b_min: "B" entries shorter than this are left out -- they go through the systolic scan, ragged_split)
void ragged_tasks(const LBAudioDetectiveCorpus* c, uint64_t nq, uint64_t b_min, uint64_t& tasks_a, uint64_t& tasks_b) {
    tasks_a = tasks_b = 0;
    for (const auto& kv : c->len_hist) {
        const uint64_t ne = kv.first;
        if (ne > nq) tasks_a += kv.second * ((ne - nq + 4) / 4);
        else if (ne >= b_min) tasks_b += kv.second * ((nq - ne + 4) / 4);
    }
}

// Split the scan?  An entry of n <= 15 sub-fingerprints against a longer query of nq costs the task kernel a pass of nq steps
// per four of its nq - n + 1 offsets, n of which meet the entry: measured 2 300 G (step, offset) slots per second whatever
// n is.  The systolic scan spends nq steps on EVERY record of a chunk that holds such an entry (2 200 G record-steps per
// second, and not less than reading the records once).  Worth a second launch when the short entries' slots are well above
// the whole corpus' record-steps.  Kernel variant 3 forces the split (where one exists), 4 forbids it.
uint32_t ragged_split(const LBAudioDetectiveCorpus* c, uint64_t nq) {
    if (sliding_short((uint32_t)nq, c->ne_max) || nq < kSlideSplitBelow || c->variant == 4) return 0;
    uint64_t slots = 0, entries = 0;
    for (const auto& kv : c->len_hist) {
        const uint64_t ne = kv.first;
        if (ne >= kSlideSplitBelow || ne > nq) continue;
        slots += kv.second * ((nq - ne + 4) / 4) * 4 * nq;
        entries += kv.second;
    }
    if (entries == 0) return 0;
    if (c->variant == 3) return kSlideSplitBelow;
    return slots > 2 * c->n_pos * nq + 20000000ull ? kSlideSplitBelow : 0;     // (+ 10 us of slots: a second launch is not free)
}

// ONE launch: the n_q queries qs[0..n_q) (all of qs[0]->count sub-fingerprints), their keys to keys + pos[i]
OSStatus launch_ragged(LBAudioDetectiveCorpus* c, const LBAudioDetectiveFingerprintRef* qs, const uint32_t* pos, uint32_t n_q,
                       uint32_t range, uint64_t index_base, float* d_scores, unsigned long long* keys, hipStream_t stream) {
    const uint32_t nq = qs[0]->count;
    const size_t block_words = ((size_t)nq + 1u) * 16u;
    std::vector<uint32_t> block, all;
    all.reserve(block_words * n_q);
    for (uint32_t i = 0; i < n_q; ++i) {
        build_sliding_query(qs[i]->data.data(), nq, c->subfp_len, range, block);
        all.insert(all.end(), block.begin() + (block.size() - block_words), block.end());      // (without the header)
    }
    const uint32_t b_min = ragged_split(c, nq);
    const bool in_args = n_q == 1 && nq <= kSlideQueryArgSubs && !sliding_short(nq, c->ne_max) && b_min == 0;   // (the systolic scan reads d_query)
    const size_t slot_words = (all.size() + 63) & ~(size_t)63;
    if (c->query_slot_words < slot_words) {               // (re)size the ring: everything that used it must be done
        for (hipEvent_t e : c->query_ev)
            if (e) LBAD_HIP(hipEventSynchronize(e));
        c->query_slot_words = 0;
        OSStatus st = reserve_query(c, slot_words * kQuerySlots);
        if (st != noErr) return st;
        c->query_slot_words = slot_words;
    }
    if (!c->d_scan_out) {
        LBAD_HIP(hipMalloc(reinterpret_cast<void**>(&c->d_scan_out), (size_t)kQuerySlots * kScanOutWords * 8));
        c->scan_out_dirty = true;
    }
    if (c->scan_out_dirty) {
        // the result words must be zero between scans: the scans themselves leave them so, but a launch that failed may not
        // have -- nothing is trusted after one: everything that may still touch the words finishes, then they are cleared
        for (hipEvent_t e : c->query_ev)
            if (e) (void)hipEventSynchronize(e);
        // on the scan's own stream and awaited: the scan may run on a non-blocking stream, which a null-stream memset does
        // not order itself against (round-5 advice)
        LBAD_HIP(hipMemsetAsync(c->d_scan_out, 0, (size_t)kQuerySlots * kScanOutWords * 8, stream));
        LBAD_HIP(hipStreamSynchronize(stream));
        c->scan_out_dirty = false;
    }
    const uint32_t slot = (uint32_t)(c->query_seq++ % kQuerySlots);
    if (!c->query_ev[slot]) LBAD_HIP(hipEventCreateWithFlags(&c->query_ev[slot], hipEventDisableTiming));
    else LBAD_HIP(hipEventSynchronize(c->query_ev[slot]));
    uint32_t* h = c->h_query + (size_t)slot * c->query_slot_words;
    uint32_t* dq = c->d_query + (size_t)slot * c->query_slot_words;
    std::memcpy(h, all.data(), all.size() * sizeof(uint32_t));
    if (!in_args) LBAD_HIP(hipMemcpyAsync(dq, h, all.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    if (d_scores) LBAD_HIP(hipMemsetAsync(d_scores, 0, c->count * sizeof(float), stream));
    uint64_t tasks_a = 0, tasks_b = 0;
    ragged_tasks(c, nq, b_min, tasks_a, tasks_b);
    if (tasks_a > 0xFFFFFFFFull || tasks_b > 0xFFFFFFFFull) return kLBAudioDetectiveArgumentInvalid;   // the plan counts in 32 bits
    const SlideShape sh = sliding_shape(tasks_a, tasks_b, n_q);
    // the plan of this query length: kept while the length and the entries stay (queries of one length are the rule)
    if (!sliding_short(nq, c->ne_max) && (c->plan_nq != nq || c->plan_count != c->count || c->plan_grid != sh.grid || c->plan_bmin != b_min)) {
        // scans on other streams may still read the old plan: every scan leaves its slot's event behind, and a slot is
        // reused only after its event -- the eight events cover everything that can still be running
        for (hipEvent_t e : c->query_ev)
            if (e) LBAD_HIP(hipEventSynchronize(e));
        if (!c->plan_built) LBAD_HIP(hipEventCreateWithFlags(&c->plan_built, hipEventDisableTiming));
        c->plan_nq = 0;
        LBAD_HIP(launch_sliding_plan(c->d_off, c->count, nq, b_min, sh, c->d_plan, stream));
        LBAD_HIP(hipEventRecord(c->plan_built, stream));
        c->plan_stream = stream;
        c->plan_nq = nq; c->plan_count = c->count; c->plan_grid = sh.grid; c->plan_bmin = b_min;
    } else if (c->plan_built && c->plan_stream != stream) {
        LBAD_HIP(hipStreamWaitEvent(stream, c->plan_built, 0));
    }
    SlideScan scan;
    scan.d_queries = in_args ? nullptr : dq;
    scan.h_query = in_args ? h : nullptr;
    scan.n_q = n_q;
    scan.d_acc = c->d_scan_out + (size_t)slot * kScanOutWords;
    scan.d_ticket = reinterpret_cast<unsigned int*>(scan.d_acc + 8);
    scan.d_keys = keys;
    for (uint32_t i = 0; i < 8; ++i) scan.key_pos[i] = i < n_q ? pos[i] : 0u;
    {
        const hipError_t launched = launch_compare_sliding(c->d_recs, c->n_pos, c->d_off, c->count, c->ne_max,
                                                           (uint32_t)(c->rec_capacity + kRecordSlack / 2), tasks_a, tasks_b, sh, c->d_plan,
                                                           c->subfp_len, scan, nq, range, index_base,
                                                           reinterpret_cast<unsigned int*>(d_scores), stream, c->bound_pruning, c->prune_from, b_min);
        if (launched != hipSuccess) c->scan_out_dirty = true;
        LBAD_HIP(launched);
    }
    // behind the SCAN, not just the copy: the slot's device half and its result words are the kernel's, and the launch
    // that reuses the slot eight launches later may arrive on another stream
    LBAD_HIP(hipEventRecord(c->query_ev[slot], stream));
    return noErr;
}

// n queries against the ragged corpus, key i to keys[i]: queries of one length share launches
OSStatus run_queries_ragged(LBAudioDetectiveCorpus* c, const LBAudioDetectiveFingerprintRef* qs, uint32_t n, uint32_t range,
                            uint64_t index_base, float* d_scores, unsigned long long* keys, hipStream_t stream) {
    if (c->count == 0) {                                   // nothing to scan: every key is "no match"
        LBAD_HIP(hipMemsetAsync(keys, 0, (size_t)n * sizeof(unsigned long long), stream));
        return noErr;
    }
    bool any_short = false;                                // the systolic scan of short queries max-es its keys in place
    for (uint32_t i = 0; i < n; ++i) any_short = any_short || sliding_short(qs[i]->count, c->ne_max) || (n > 1 && sliding_multi(qs[i]->count, c->ne_max));
    if (any_short) LBAD_HIP(hipMemsetAsync(keys, 0, (size_t)n * sizeof(unsigned long long), stream));
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return qs[x]->count < qs[y]->count; });
    for (uint32_t at = 0; at < n;) {
        uint32_t same = 1;
        while (at + same < n && qs[order[at + same]]->count == qs[order[at]]->count) ++same;
        while (same) {
            const uint32_t g = d_scores ? 1u : sliding_queries_per_launch(qs[order[at]]->count, c->ne_max, same);
            LBAudioDetectiveFingerprintRef group[8];
            uint32_t pos[8];
            for (uint32_t i = 0; i < g; ++i) { group[i] = qs[order[at + i]]; pos[i] = order[at + i]; }
            OSStatus st = launch_ragged(c, group, pos, g, range, index_base, d_scores, keys, stream);
            if (st != noErr) return st;
            at += g;
            same -= g;
        }
    }
    return noErr;
}

OSStatus run_query_ragged(LBAudioDetectiveCorpus* c, const LBAudioDetectiveFingerprint* q, uint32_t range,
                          uint64_t index_base, float* d_scores, unsigned long long* key_dst, hipStream_t stream) {
    LBAudioDetectiveFingerprintRef one = const_cast<LBAudioDetectiveFingerprint*>(q);
    return run_queries_ragged(c, &one, 1, range, index_base, d_scores, key_dst, stream);
}

// stage the query on the device and launch the scan; key_dst is a device pointer
OSStatus run_query_impl(LBAudioDetectiveCorpus* c, const LBAudioDetectiveFingerprint* q, uint32_t range,
                   uint64_t index_base, float* d_scores, unsigned long long* key_dst, hipStream_t stream) {
    if (!c || !q || !key_dst) return kLBAudioDetectiveArgumentInvalid;
    if (q->length != c->subfp_len || q->count == 0) return kLBAudioDetectiveArgumentInvalid;
    if (!c->ragged && (size_t)q->count * kPackedWords * 4 > 48 * 1024) return kLBAudioDetectiveArgumentInvalid;
    if (range == 0) range = c->subfp_len;  // LBAudioDetective.m:443-445
    if (c->ragged) return run_query_ragged(c, q, range, index_base, d_scores, key_dst, stream);
    std::vector<uint32_t> slots;
    pack_fingerprint(q, slots);
    bool fast = planes_fast_supported(c->subfp_len, c->n_sub, q->count);
    if (c->variant == 1) fast = false;
    if (c->variant == 2 && !fast) return kLBAudioDetectiveArgumentInvalid;
    LBAD_HIP(hipMemsetAsync(key_dst, 0, sizeof(unsigned long long), stream));
    if (fast) {
        // the specialised scan takes its 300-byte query block as a kernel argument: nothing to stage
        std::vector<uint32_t> block;
        build_plane_query(slots.data(), c->n_sub, range, block);
        LBAD_HIP(launch_compare_planes_fast(c->d_planes, c->capacity, c->count, c->n_sub, block.data(), index_base,
                                            d_scores, key_dst, stream));
        return noErr;
    }
    if (c->query_cap < slots.size()) {
        if (c->query_ev[0]) LBAD_HIP(hipEventSynchronize(c->query_ev[0]));
        if (c->d_query) (void)hipFree(c->d_query);
        if (c->h_query) (void)hipHostFree(c->h_query);
        c->d_query = nullptr;
        c->h_query = nullptr;
        c->query_cap = 0;
        LBAD_HIP(hipMalloc(reinterpret_cast<void**>(&c->d_query), slots.size() * sizeof(uint32_t)));
        LBAD_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_query), slots.size() * sizeof(uint32_t), hipHostMallocDefault));
        c->query_cap = (uint32_t)slots.size();
    }
    // the staging block and its device copy are reused by every query: wait for the previous one's SCAN (whatever
    // stream it ran on; slot 0 of the ragged ring's events serves this path, a corpus is either ragged or not)
    if (!c->query_ev[0]) LBAD_HIP(hipEventCreateWithFlags(&c->query_ev[0], hipEventDisableTiming));
    else LBAD_HIP(hipEventSynchronize(c->query_ev[0]));
    std::memcpy(c->h_query, slots.data(), slots.size() * sizeof(uint32_t));
    LBAD_HIP(hipMemcpyAsync(c->d_query, c->h_query, slots.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    LBAD_HIP(launch_compare_planes_generic(c->d_planes, c->capacity, c->count, c->n_sub, c->subfp_len, c->d_query,
                                           q->count, range, index_base, d_scores, key_dst, stream));
    LBAD_HIP(hipEventRecord(c->query_ev[0], stream));
    return noErr;
}

// (host allocations -- the packed query, its constant block -- can fail: nothing may unwind through the C boundary)
OSStatus run_query(LBAudioDetectiveCorpus* c, const LBAudioDetectiveFingerprint* q, uint32_t range, uint64_t index_base,
                   float* d_scores, unsigned long long* key_dst, hipStream_t stream) {
    LBAD_GUARD_BEGIN
    return run_query_impl(c, q, range, index_base, d_scores, key_dst, stream);
    LBAD_GUARD_END
}

// LBAudioDetectiveCorpusQuery on the specialised scan: one launch, the result arrives in pinned memory
OSStatus query_fast_impl(LBAudioDetectiveCorpus* c, const LBAudioDetectiveFingerprint* q, uint32_t range, unsigned long long* key) {
    if (!c->h_out) {
        // built in locals and committed only when everything exists: a failure leaves the corpus as it was
        unsigned long long *d_fast = nullptr, *h_out = nullptr, *h_out_dev = nullptr;
        OSStatus st = c->stream ? noErr : kLBAudioDetectiveDeviceError;     // created with the corpus
        if (st == noErr) st = hip_status(hipMalloc(reinterpret_cast<void**>(&d_fast), (kScanSlots + 1) * sizeof(unsigned long long)), "hipMalloc", __LINE__);
        if (st == noErr) st = hip_status(hipMemset(d_fast, 0, (kScanSlots + 1) * sizeof(unsigned long long)), "memset", __LINE__);
        if (st == noErr) st = hip_status(hipHostMalloc(reinterpret_cast<void**>(&h_out), 16, hipHostMallocMapped | hipHostMallocCoherent), "hipHostMalloc", __LINE__);
        if (st == noErr) {
            h_out[0] = h_out[1] = 0;
            st = hip_status(hipHostGetDevicePointer(reinterpret_cast<void**>(&h_out_dev), h_out, 0), "device pointer", __LINE__);
        }
        if (st != noErr || !h_out_dev) {
            if (h_out) (void)hipHostFree(h_out);
            if (d_fast) (void)hipFree(d_fast);
            return st != noErr ? st : kLBAudioDetectiveDeviceError;
        }
        c->d_fast_key = d_fast;
        c->d_ticket = reinterpret_cast<unsigned int*>(d_fast + kScanSlots);
        c->h_out_dev = h_out_dev;
        c->h_out = h_out;
    }
    if (range == 0) range = c->subfp_len;
    std::vector<uint32_t> slots, block;
    pack_fingerprint(q, slots);
    build_plane_query(slots.data(), c->n_sub, range, block);
    const unsigned long long seq = ++c->seq;
    LBAD_HIP(launch_compare_planes_fast(c->d_planes, c->capacity, c->count, c->n_sub, block.data(), 0, nullptr,
                                        c->d_fast_key, c->stream, c->d_ticket, c->h_out_dev, seq));
    volatile unsigned long long* out = c->h_out;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spins = 1; out[1] != seq; ++spins) {
        if ((spins & 0xFFFFF) == 0) {                       // every million polls: is the stream still alive?
            const hipError_t e = hipStreamQuery(c->stream);
            if (e == hipSuccess) {                          // the kernel is done: its words are on their way or lost
                LBAD_HIP(hipStreamSynchronize(c->stream));
                if (out[1] != seq) return kLBAudioDetectiveDeviceError;
            } else if (e != hipErrorNotReady) {
                return hip_status(e, "corpus query", __LINE__);
            } else if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) {
                fprintf(stderr, "lbaudiodetective: corpus scan not finished after 30 s\n");   // a hung kernel must not hang the host
                return kLBAudioDetectiveDeviceError;
            }
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    *key = out[0];
    return noErr;
}

OSStatus query_fast(LBAudioDetectiveCorpus* c, const LBAudioDetectiveFingerprint* q, uint32_t range, unsigned long long* key) {
    LBAD_GUARD_BEGIN
    return query_fast_impl(c, q, range, key);
    LBAD_GUARD_END
}

}  // namespace
}  // namespace lbad

extern "C" {

LBAudioDetectiveCorpusRef LBAudioDetectiveCorpusNew(UInt32 inSubfingerprintLength, UInt32 inSubfingerprintsPerEntry,
                                                    UInt64 inCapacity) {
    if (inCapacity == 0 || inCapacity > 0xFFFFFFFFull) return NULL;  // the key carries a 32-bit index
    if (!lbad::planes_supported(inSubfingerprintLength, inSubfingerprintsPerEntry)) return NULL;
    if (!lbad::device_ready()) {
        fprintf(stderr, "lbaudiodetective: no HIP device, cannot create a corpus\n");
        return NULL;
    }
    LBAudioDetectiveCorpus* c = new LBAudioDetectiveCorpus();
    c->subfp_len = inSubfingerprintLength;
    c->n_sub = inSubfingerprintsPerEntry;
    c->capacity = inCapacity;
    c->n_planes = lbad::planes_per_entry(inSubfingerprintLength, inSubfingerprintsPerEntry);
    const size_t bytes = (size_t)c->n_planes * inCapacity * sizeof(uint4);
    // the polled query's own stream exists from the start, so that every append can order it behind itself
    if (lbad::hip_status(hipMalloc(reinterpret_cast<void**>(&c->d_planes), bytes), "hipMalloc corpus", __LINE__) != noErr ||
        lbad::hip_status(hipMalloc(reinterpret_cast<void**>(&c->d_key), 16), "hipMalloc key", __LINE__) != noErr ||
        lbad::hip_status(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), "stream", __LINE__) != noErr ||
        lbad::hip_status(hipMalloc(reinterpret_cast<void**>(&c->d_shard_keys), (size_t)LBAD_SHARD_KEYS * 8), "hipMalloc keys", __LINE__) != noErr ||
        lbad::hip_status(hipHostMalloc(reinterpret_cast<void**>(&c->h_shard_keys), (size_t)LBAD_SHARD_KEYS * 8, hipHostMallocDefault), "keys", __LINE__) != noErr) {
        LBAudioDetectiveCorpusDispose(c);
        return NULL;
    }
    return c;
}

LBAudioDetectiveCorpusRef LBAudioDetectiveCorpusNewRagged(UInt32 inSubfingerprintLength, UInt64 inEntryCapacity,
                                                          UInt64 inSubfingerprintCapacity) {
    if (inEntryCapacity == 0 || inEntryCapacity > lbad::kMaxRaggedEntries) return NULL;   // the key carries a 32-bit index (and the scan's claim cursor a little slack)
    if (inSubfingerprintCapacity < inEntryCapacity || inSubfingerprintCapacity > lbad::kMaxRaggedRecords) return NULL;
    if (!lbad::sliding_supported(inSubfingerprintLength)) return NULL;
    if (!lbad::device_ready()) {
        fprintf(stderr, "lbaudiodetective: no HIP device, cannot create a corpus\n");
        return NULL;
    }
    LBAudioDetectiveCorpus* c = new (std::nothrow) LBAudioDetectiveCorpus();
    if (!c) return NULL;
    c->ragged = true;
    c->subfp_len = inSubfingerprintLength;
    c->capacity = inEntryCapacity;
    c->rec_capacity = inSubfingerprintCapacity;
    try {
        c->h_off.assign(1, 0u);
    } catch (const std::bad_alloc&) {
        delete c;
        return NULL;
    }
    // kRecordSlack zero records behind the capacity: the scan reads up to three records past an entry's end (offsets
    // that do not exist, never used) and takes its all-zero record from there
    if (lbad::hip_status(hipMalloc(reinterpret_cast<void**>(&c->d_recs), ((size_t)inSubfingerprintCapacity + lbad::kRecordSlack) * 32), "hipMalloc corpus", __LINE__) != noErr ||
        lbad::hip_status(hipMemset(c->d_recs + 2 * (size_t)inSubfingerprintCapacity, 0, (size_t)lbad::kRecordSlack * 32), "corpus slack", __LINE__) != noErr ||
        lbad::hip_status(hipMalloc(reinterpret_cast<void**>(&c->d_off), (size_t)(inEntryCapacity + 1) * 4), "hipMalloc offsets", __LINE__) != noErr ||
        lbad::hip_status(hipMemset(c->d_off, 0, 4), "offsets", __LINE__) != noErr ||
        lbad::hip_status(hipMalloc(reinterpret_cast<void**>(&c->d_key), 16), "hipMalloc key", __LINE__) != noErr ||
        lbad::hip_status(hipMalloc(reinterpret_cast<void**>(&c->d_plan), lbad::sliding_plan_words(inEntryCapacity) * 4), "hipMalloc plan", __LINE__) != noErr ||
        lbad::hip_status(hipMalloc(reinterpret_cast<void**>(&c->d_shard_keys), (size_t)LBAD_SHARD_KEYS * 8), "hipMalloc keys", __LINE__) != noErr ||
        lbad::hip_status(hipHostMalloc(reinterpret_cast<void**>(&c->h_shard_keys), (size_t)LBAD_SHARD_KEYS * 8, hipHostMallocDefault), "keys", __LINE__) != noErr) {
        LBAudioDetectiveCorpusDispose(c);
        return NULL;
    }
    return c;
}

UInt64 LBAudioDetectiveCorpusGetSubfingerprintTotal(LBAudioDetectiveCorpusRef c) {
    if (!c) return 0;
    return c->ragged ? c->n_pos : c->count * c->n_sub;
}

OSStatus LBAudioDetectiveCorpusAppendRaggedPackedDevice(LBAudioDetectiveCorpusRef c, const void* inPacked,
                                                        const UInt32* inCounts, UInt64 inNumberOfEntries, void* inStream) {
    if (!c || !c->ragged || (inNumberOfEntries && (!inPacked || !inCounts))) return kLBAudioDetectiveArgumentInvalid;
    if (inNumberOfEntries == 0) return noErr;
    if (c->count + inNumberOfEntries > c->capacity) return kLBAudioDetectiveArgumentInvalid;
    LBAD_GUARD_BEGIN
    uint64_t total = 0;
    uint32_t longest = c->ne_max;
    for (UInt64 e = 0; e < inNumberOfEntries; ++e) {
        if (inCounts[e] == 0) return kLBAudioDetectiveArgumentInvalid;   // an entry has at least one sub-fingerprint
        total += inCounts[e];
        if (inCounts[e] > longest) longest = inCounts[e];
    }
    if (c->n_pos + total > c->rec_capacity) return kLBAudioDetectiveArgumentInvalid;
    const size_t at = c->h_off.size() - 1;            // == c->count
    c->h_off.resize(at + inNumberOfEntries + 1);
    for (UInt64 e = 0; e < inNumberOfEntries; ++e) c->h_off[at + e + 1] = c->h_off[at + e] + inCounts[e];
    hipStream_t stream = static_cast<hipStream_t>(inStream);
    OSStatus st = lbad::hip_status(hipMemcpyAsync(c->d_off + at, c->h_off.data() + at, (inNumberOfEntries + 1) * 4,
                                                  hipMemcpyHostToDevice, stream), "offsets H2D", __LINE__);
    if (st == noErr)
        st = lbad::hip_status(lbad::launch_pack_records(static_cast<const uint32_t*>(inPacked), total, c->d_off + at,
                                                        inNumberOfEntries, (uint32_t)c->count, c->d_recs, stream),
                              "pack records", __LINE__);
    if (st == noErr) {
        try {
            for (UInt64 e = 0; e < inNumberOfEntries; ++e) ++c->len_hist[inCounts[e]];
        } catch (const std::bad_alloc&) {
            // the entries up to e are counted: rebuild the histogram from the offsets that stay
            c->h_off.resize(at + 1);
            c->len_hist.clear();
            for (size_t k = 0; k + 1 < c->h_off.size(); ++k) ++c->len_hist[c->h_off[k + 1] - c->h_off[k]];
            return kLBAudioDetectiveMemFull;
        }
    }
    if (st != noErr) {
        c->h_off.resize(at + 1);
        return st;
    }
    c->count += inNumberOfEntries;
    c->n_pos += total;
    c->ne_max = longest;
    return noErr;
    LBAD_GUARD_END
}

void LBAudioDetectiveCorpusDispose(LBAudioDetectiveCorpusRef c) {
    if (!c) return;
    if (c->plan_built) { (void)hipEventSynchronize(c->plan_built); (void)hipEventDestroy(c->plan_built); }
    if (c->d_plan) (void)hipFree(c->d_plan);
    if (c->d_scan_out) (void)hipFree(c->d_scan_out);
    if (c->d_shard_keys) (void)hipFree(c->d_shard_keys);
    if (c->h_shard_keys) (void)hipHostFree(c->h_shard_keys);
    if (c->d_recs) (void)hipFree(c->d_recs);
    if (c->d_off) (void)hipFree(c->d_off);
    if (c->d_planes) (void)hipFree(c->d_planes);
    for (hipEvent_t e : c->query_ev)
        if (e) { (void)hipEventSynchronize(e); (void)hipEventDestroy(e); }
    if (c->d_query) (void)hipFree(c->d_query);
    if (c->h_query) (void)hipHostFree(c->h_query);
    if (c->d_key) (void)hipFree(c->d_key);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->append_event) (void)hipEventDestroy(c->append_event);
    if (c->shard_stale_event) (void)hipEventDestroy(c->shard_stale_event);
    if (c->d_fast_key) (void)hipFree(c->d_fast_key);
    if (c->h_out) (void)hipHostFree(c->h_out);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

UInt64 LBAudioDetectiveCorpusGetCount(LBAudioDetectiveCorpusRef c) { return c ? c->count : 0; }
unsigned long long* LBAudioDetectiveCorpusShardKeysDevice(LBAudioDetectiveCorpusRef c) { return c ? c->d_shard_keys : NULL; }
unsigned long long* LBAudioDetectiveCorpusShardKeysHost(LBAudioDetectiveCorpusRef c) { return c ? c->h_shard_keys : NULL; }

UInt32 LBAudioDetectiveCorpusGetEntryStrideBytes(LBAudioDetectiveCorpusRef c) {
    if (!c) return 0;
    return c->ragged ? 32u : c->n_planes * 16u;   // ragged: bytes per sub-fingerprint record
}

OSStatus LBAudioDetectiveCorpusSetKernelVariant(LBAudioDetectiveCorpusRef c, UInt32 inVariant) {
    if (!c || inVariant > 4 || (inVariant > 2 && !c->ragged)) return kLBAudioDetectiveArgumentInvalid;
    c->variant = inVariant;
    return noErr;
}

OSStatus LBAudioDetectiveCorpusAppendPackedDevice(LBAudioDetectiveCorpusRef c, const void* inPacked,
                                                  UInt64 inNumberOfEntries, void* inStream) {
    if (!c || c->ragged || (!inPacked && inNumberOfEntries)) return kLBAudioDetectiveArgumentInvalid;
    if (c->count + inNumberOfEntries > c->capacity) return kLBAudioDetectiveArgumentInvalid;
    LBAD_HIP(lbad::launch_pack_planes(static_cast<const uint32_t*>(inPacked), inNumberOfEntries, c->n_sub, c->subfp_len,
                                      c->d_planes, c->capacity, c->count, static_cast<hipStream_t>(inStream)));
    // the polled query runs on the corpus's own stream: it waits for this event instead of keeping the caller's
    // stream handle (which may be gone by then); appends on several streams each leave their event behind
    if (!c->append_event) LBAD_HIP(hipEventCreateWithFlags(&c->append_event, hipEventDisableTiming));
    LBAD_HIP(hipEventRecord(c->append_event, static_cast<hipStream_t>(inStream)));
    if (c->stream) LBAD_HIP(hipStreamWaitEvent(c->stream, c->append_event, 0));
    c->appended = true;
    c->count += inNumberOfEntries;
    return noErr;
}

OSStatus LBAudioDetectiveCorpusAppendFingerprint(LBAudioDetectiveCorpusRef c, LBAudioDetectiveFingerprintRef fp) {
    LBAD_GUARD_BEGIN
    if (!c || !fp || fp->length != c->subfp_len) return kLBAudioDetectiveArgumentInvalid;
    if (c->ragged ? fp->count == 0 : fp->count != c->n_sub) return kLBAudioDetectiveArgumentInvalid;
    std::vector<uint32_t> slots;
    lbad::pack_fingerprint(fp, slots);
    uint32_t* d = nullptr;
    LBAD_HIP(hipMalloc(reinterpret_cast<void**>(&d), slots.size() * 4));
    OSStatus st = lbad::hip_status(hipMemcpy(d, slots.data(), slots.size() * 4, hipMemcpyHostToDevice), "copy", __LINE__);
    const UInt32 n = fp->count;
    if (st == noErr) st = c->ragged ? LBAudioDetectiveCorpusAppendRaggedPackedDevice(c, d, &n, 1, NULL)
                                    : LBAudioDetectiveCorpusAppendPackedDevice(c, d, 1, NULL);
    if (st == noErr) st = lbad::hip_status(hipStreamSynchronize(nullptr), "sync", __LINE__);
    (void)hipFree(d);
    return st;
    LBAD_GUARD_END
}

void LBAudioDetectiveCorpusDecodeKey(UInt64 inKey, SInt64* outIndex, Float32* outScore) {
    const uint32_t bits = (uint32_t)(inKey >> 32);
    float score;
    std::memcpy(&score, &bits, 4);
    if (outScore) *outScore = score;
    // strict '<' against an initial 0.0 (LBAudioDetectiveTests.m:60,80): a best score of 0 selects nothing
    if (outIndex) *outIndex = (inKey == 0 || !(score > 0.0f)) ? -1 : (SInt64)(0xFFFFFFFFu - (uint32_t)inKey);
}

OSStatus LBAudioDetectiveCorpusQueryKeyDevice(LBAudioDetectiveCorpusRef c, LBAudioDetectiveFingerprintRef inQuery,
                                              UInt32 inRange, UInt64 inIndexBase, void* outKey, void* inStream) {
    if (c && inIndexBase + c->count > 0x100000000ull) return kLBAudioDetectiveArgumentInvalid;
    return lbad::run_query(c, inQuery, inRange, inIndexBase, nullptr, static_cast<unsigned long long*>(outKey),
                           static_cast<hipStream_t>(inStream));
}

OSStatus LBAudioDetectiveCorpusScoresDevice(LBAudioDetectiveCorpusRef c, LBAudioDetectiveFingerprintRef inQuery,
                                            UInt32 inRange, Float32* outScores, void* inStream) {
    if (!c || !outScores) return kLBAudioDetectiveArgumentInvalid;
    return lbad::run_query(c, inQuery, inRange, 0, outScores, c->d_key, static_cast<hipStream_t>(inStream));
}

// Several queries in one pass over the corpus.  outKeys is a device pointer to inCount 64-bit keys.
OSStatus LBAudioDetectiveCorpusQueryBatchKeysDevice(LBAudioDetectiveCorpusRef c, const LBAudioDetectiveFingerprintRef* inQueries,
                                                    UInt32 inCount, UInt32 inRange, UInt64 inIndexBase, void* outKeys,
                                                    void* inStream) {
    LBAD_GUARD_BEGIN
    if (!c || !inQueries || !outKeys || inCount == 0) return kLBAudioDetectiveArgumentInvalid;
    if (inIndexBase + c->count > 0x100000000ull) return kLBAudioDetectiveArgumentInvalid;
    hipStream_t stream = static_cast<hipStream_t>(inStream);
    unsigned long long* keys = static_cast<unsigned long long*>(outKeys);
    bool all_fast = c->variant != 1;
    for (UInt32 i = 0; i < inCount && all_fast; ++i)
        all_fast = inQueries[i] && lbad::planes_fast_supported(c->subfp_len, c->n_sub, inQueries[i]->count) &&
                   inQueries[i]->length == c->subfp_len;
    if (c->ragged) {   // queries of one length share their passes over the records (k_sliding.hip)
        for (UInt32 i = 0; i < inCount; ++i)
            if (!inQueries[i] || inQueries[i]->length != c->subfp_len || inQueries[i]->count == 0) return kLBAudioDetectiveArgumentInvalid;
        return lbad::run_queries_ragged(c, inQueries, inCount, inRange ? inRange : c->subfp_len, inIndexBase, nullptr, keys, stream);
    }
    if (!all_fast) {   // shapes without the specialised scan: one pass per query
        for (UInt32 i = 0; i < inCount; ++i) {
            OSStatus st = lbad::run_query(c, inQueries[i], inRange, inIndexBase, nullptr, keys + i, stream);
            if (st != noErr) return st;
        }
        return noErr;
    }
    const uint32_t range = inRange ? inRange : c->subfp_len;
    const uint32_t kw = lbad::plane_query_words();
    const size_t words = (size_t)inCount * kw;
    // the staging pair is shared with the single-query generic scan, which may still be running on ANOTHER stream:
    // its event (slot 0) orders every reuse, this path's too (round-3 advice)
    if (c->query_ev[0]) LBAD_HIP(hipEventSynchronize(c->query_ev[0]));
    if (c->query_cap < words) {
        if (c->d_query) (void)hipFree(c->d_query);
        if (c->h_query) (void)hipHostFree(c->h_query);
        c->d_query = nullptr;
        c->h_query = nullptr;
        c->query_cap = 0;
        LBAD_HIP(hipMalloc(reinterpret_cast<void**>(&c->d_query), words * sizeof(uint32_t)));
        LBAD_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_query), words * sizeof(uint32_t), hipHostMallocDefault));
        c->query_cap = (uint32_t)words;
    }
    LBAD_HIP(hipStreamSynchronize(stream));   // the pinned staging block is reused by every call
    std::memset(c->h_query, 0, words * sizeof(uint32_t));
    std::vector<uint32_t> slots, block;
    for (UInt32 i = 0; i < inCount; ++i) {
        lbad::pack_fingerprint(inQueries[i], slots);
        lbad::build_plane_query(slots.data(), c->n_sub, range, block);
        std::memcpy(c->h_query + (size_t)i * kw, block.data(), block.size() * sizeof(uint32_t));
    }
    LBAD_HIP(hipMemcpyAsync(c->d_query, c->h_query, words * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    LBAD_HIP(hipMemsetAsync(keys, 0, (size_t)inCount * sizeof(unsigned long long), stream));
    LBAD_HIP(lbad::launch_compare_planes_batch(c->d_planes, c->capacity, c->count, c->n_sub, c->d_query, inCount,
                                               inIndexBase, keys, stream));
    if (!c->query_ev[0]) LBAD_HIP(hipEventCreateWithFlags(&c->query_ev[0], hipEventDisableTiming));
    LBAD_HIP(hipEventRecord(c->query_ev[0], stream));
    return noErr;
    LBAD_GUARD_END
}

OSStatus LBAudioDetectiveCorpusQueryBatch(LBAudioDetectiveCorpusRef c, const LBAudioDetectiveFingerprintRef* inQueries,
                                          UInt32 inCount, UInt32 inRange, SInt64* outIndices, Float32* outScores) {
    LBAD_GUARD_BEGIN
    if (!c || inCount == 0) return kLBAudioDetectiveArgumentInvalid;
    std::vector<unsigned long long> keys(inCount);
    unsigned long long* d_keys = nullptr;
    LBAD_HIP(hipMalloc(reinterpret_cast<void**>(&d_keys), (size_t)inCount * sizeof(unsigned long long)));
    OSStatus st = LBAudioDetectiveCorpusQueryBatchKeysDevice(c, inQueries, inCount, inRange, 0, d_keys, NULL);
    if (st == noErr)
        st = lbad::hip_status(hipMemcpy(keys.data(), d_keys, keys.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost),
                              "copy keys", __LINE__);
    (void)hipFree(d_keys);
    if (st != noErr) return st;
    for (UInt32 i = 0; i < inCount; ++i)
        LBAudioDetectiveCorpusDecodeKey(keys[i], outIndices ? outIndices + i : NULL, outScores ? outScores + i : NULL);
    return noErr;
    LBAD_GUARD_END
}

// ---- corpus file: header + the planes of the stored entries, plane-major ----------------------------
namespace {
struct CorpusFileHeader {
    char magic[8];            // "LBADCRP1"
    uint32_t subfp_len, n_sub, n_planes, reserved;
    uint64_t count;
};
}  // namespace

// ragged corpus file: header, the entries' sub-fingerprint counts, the records
namespace {
struct RaggedFileHeader {
    char magic[8];            // "LBADCRP3" (round-4 record layout, k_sliding.hip); "LBADCRP2" files (round 3) still load
    uint32_t subfp_len, reserved;
    uint64_t count, n_pos;
};

OSStatus save_ragged(LBAudioDetectiveCorpus* c, FILE* f) {
    RaggedFileHeader h;
    std::memcpy(h.magic, "LBADCRP3", 8);
    h.subfp_len = c->subfp_len; h.reserved = 0; h.count = c->count; h.n_pos = c->n_pos;
    if (std::fwrite(&h, sizeof(h), 1, f) != 1) return kLBAudioDetectiveDeviceError;
    std::vector<uint32_t> counts(c->count);
    for (uint64_t e = 0; e < c->count; ++e) counts[e] = c->h_off[e + 1] - c->h_off[e];
    if (c->count && std::fwrite(counts.data(), 4, c->count, f) != c->count) return kLBAudioDetectiveDeviceError;
    const size_t piece = 1u << 20;                        // records per staging piece (32 MiB)
    std::vector<uint4> host(2 * (c->n_pos < piece ? (size_t)c->n_pos : piece));
    for (uint64_t at = 0; at < c->n_pos; at += piece) {
        const size_t n = (size_t)(c->n_pos - at < piece ? c->n_pos - at : piece);
        OSStatus st = lbad::hip_status(hipMemcpy(host.data(), c->d_recs + 2 * at, n * 32, hipMemcpyDeviceToHost),
                                       "corpus records D2H", __LINE__);
        if (st != noErr) return st;
        if (std::fwrite(host.data(), 32, n, f) != n) return kLBAudioDetectiveDeviceError;
    }
    return noErr;
}

LBAudioDetectiveCorpusRef load_ragged(FILE* f, long file_size, uint64_t capacity) {
    RaggedFileHeader h;
    if (file_size < (long)sizeof(h) || std::fread(&h, sizeof(h), 1, f) != 1) return NULL;
    const bool old_layout = std::memcmp(h.magic, "LBADCRP2", 8) == 0;
    // untrusted header: the file must really hold what it announces before anything is allocated from it
    if (!lbad::sliding_supported(h.subfp_len) || h.count > lbad::kMaxRaggedEntries || h.n_pos > lbad::kMaxRaggedRecords || h.n_pos < h.count)
        return NULL;
    if ((uint64_t)(file_size - (long)sizeof(h)) < h.count * 4 + h.n_pos * 32) return NULL;
    std::vector<uint32_t> counts;
    try { counts.resize(h.count); } catch (const std::bad_alloc&) { return NULL; }
    if (h.count && std::fread(counts.data(), 4, h.count, f) != h.count) return NULL;
    uint64_t total = 0;
    for (uint64_t e = 0; e < h.count; ++e) {
        if (counts[e] == 0) return NULL;
        total += counts[e];
    }
    if (total != h.n_pos) return NULL;
    uint64_t cap = capacity > h.count ? capacity : (h.count ? h.count : 1);
    if (cap > lbad::kMaxRaggedEntries) cap = lbad::kMaxRaggedEntries;           // (NewRagged's own limits: a large request is clamped, not refused)
    // room for records in proportion to the entry capacity
    uint64_t rec_cap = h.count ? (uint64_t)(((unsigned __int128)h.n_pos * cap + h.count - 1) / h.count) : cap * 64;
    if (rec_cap < cap) rec_cap = cap;
    if (rec_cap > lbad::kMaxRaggedRecords) rec_cap = lbad::kMaxRaggedRecords;
    LBAudioDetectiveCorpusRef c = LBAudioDetectiveCorpusNewRagged(h.subfp_len, cap, rec_cap);
    if (!c) return NULL;
    bool ok = true;
    const size_t piece = 1u << 20;
    uint4* host = static_cast<uint4*>(std::malloc(32 * (h.n_pos < piece ? (size_t)(h.n_pos ? h.n_pos : 1) : piece)));
    if (!host) ok = false;
    for (uint64_t at = 0; ok && at < h.n_pos; at += piece) {
        const size_t n = (size_t)(h.n_pos - at < piece ? h.n_pos - at : piece);
        ok = std::fread(host, 32, n, f) == n &&
             lbad::hip_status(hipMemcpy(c->d_recs + 2 * at, host, n * 32, hipMemcpyHostToDevice), "corpus records H2D",
                              __LINE__) == noErr;
    }
    std::free(host);
    if (ok) {
        try {
            c->h_off.resize(h.count + 1);
        } catch (const std::bad_alloc&) { ok = false; }
    }
    if (ok) {
        c->h_off[0] = 0;
        try {
            for (uint64_t e = 0; e < h.count; ++e) {
                c->h_off[e + 1] = c->h_off[e] + counts[e];
                if (counts[e] > c->ne_max) c->ne_max = counts[e];
                ++c->len_hist[counts[e]];
            }
        } catch (const std::bad_alloc&) { ok = false; }
    }
    if (ok)
        ok = lbad::hip_status(hipMemcpy(c->d_off, c->h_off.data(), (h.count + 1) * 4, hipMemcpyHostToDevice),
                              "corpus offsets H2D", __LINE__) == noErr;
    // The records of a file are data, not structure: every place a record has inside its entry comes from the counts
    // validated above (the scan reads d_off, nothing else), and what a record carries besides its 200 Booleans -- the
    // table row of its `possible`, reserved bits, pairs beyond the length -- is recomputed / cleared here, so a crafted
    // file cannot steer the scan (round-3 advice: the old records carried their own index fields)
    if (ok)
        ok = lbad::hip_status(lbad::launch_restamp_records(c->d_recs, c->d_off, h.count, h.n_pos, h.subfp_len, old_layout, nullptr),
                              "restamp records", __LINE__) == noErr &&
             lbad::hip_status(hipStreamSynchronize(nullptr), "restamp records", __LINE__) == noErr;
    if (!ok) {
        LBAudioDetectiveCorpusDispose(c);
        return NULL;
    }
    c->count = h.count;
    c->n_pos = h.n_pos;
    return c;
}
}  // namespace

OSStatus LBAudioDetectiveCorpusSave(LBAudioDetectiveCorpusRef c, const char* inPath) {
    if (!c || !inPath) return kLBAudioDetectiveArgumentInvalid;
    FILE* f = std::fopen(inPath, "wb");
    if (!f) return -43;
    if (c->ragged) {
        OSStatus rst = kLBAudioDetectiveMemFull;
        try { rst = save_ragged(c, f); } catch (const std::bad_alloc&) {}
        std::fclose(f);
        return rst;
    }
    CorpusFileHeader h;
    std::memcpy(h.magic, "LBADCRP1", 8);
    h.subfp_len = c->subfp_len; h.n_sub = c->n_sub; h.n_planes = c->n_planes; h.reserved = 0; h.count = c->count;
    OSStatus st = std::fwrite(&h, sizeof(h), 1, f) == 1 ? noErr : kLBAudioDetectiveDeviceError;
    std::vector<uint4> host;
    try {
        host.resize(c->count);
    } catch (const std::bad_alloc&) {
        std::fclose(f);
        return kLBAudioDetectiveMemFull;
    }
    for (uint32_t p = 0; p < c->n_planes && st == noErr && c->count; ++p) {
        st = lbad::hip_status(hipMemcpy(host.data(), c->d_planes + (size_t)p * c->capacity, c->count * sizeof(uint4),
                                        hipMemcpyDeviceToHost), "corpus plane D2H", __LINE__);
        if (st == noErr && std::fwrite(host.data(), sizeof(uint4), c->count, f) != c->count) st = kLBAudioDetectiveDeviceError;
    }
    std::fclose(f);
    return st;
}

LBAudioDetectiveCorpusRef LBAudioDetectiveCorpusLoad(const char* inPath, UInt64 inCapacity) {
    if (!inPath) return NULL;
    FILE* f = std::fopen(inPath, "rb");
    if (!f) return NULL;
    CorpusFileHeader h;
    LBAudioDetectiveCorpusRef c = NULL;
    // the header is untrusted: the shape must be one the device path supports and the file must really hold
    // count * n_planes planes before anything is allocated from it
    std::fseek(f, 0, SEEK_END);
    const long file_size = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    char magic[8] = {0};
    if (file_size >= 8 && std::fread(magic, 8, 1, f) == 1 &&
        (std::memcmp(magic, "LBADCRP3", 8) == 0 || std::memcmp(magic, "LBADCRP2", 8) == 0)) {
        std::fseek(f, 0, SEEK_SET);
        c = load_ragged(f, file_size, inCapacity);
        std::fclose(f);
        return c;
    }
    std::fseek(f, 0, SEEK_SET);
    bool ok = file_size >= (long)sizeof(h) && std::fread(&h, sizeof(h), 1, f) == 1 &&
              std::memcmp(h.magic, "LBADCRP1", 8) == 0 && h.subfp_len > 0 &&
              h.subfp_len <= LBAD_MAX_SUBFINGERPRINT_LENGTH && h.n_sub > 0 &&
              lbad::planes_supported(h.subfp_len, h.n_sub) && h.n_planes == lbad::planes_per_entry(h.subfp_len, h.n_sub) &&
              h.count <= 0xFFFFFFFFull &&
              (uint64_t)(file_size - (long)sizeof(h)) / sizeof(uint4) / (h.n_planes ? h.n_planes : 1) >= h.count;
    if (ok) {
        const uint64_t cap = inCapacity > h.count ? inCapacity : (h.count ? h.count : 1);
        c = LBAudioDetectiveCorpusNew(h.subfp_len, h.n_sub, cap);
    }
    if (c && h.count) {
        // staged in bounded pieces, not one count-sized vector
        const size_t piece = 1u << 20;
        uint4* host = static_cast<uint4*>(std::malloc(sizeof(uint4) * (h.count < piece ? (size_t)h.count : piece)));
        if (!host) ok = false;
        for (uint32_t p = 0; ok && p < h.n_planes; ++p) {
            for (uint64_t at = 0; ok && at < h.count; at += piece) {
                const size_t n = (size_t)(h.count - at < piece ? h.count - at : piece);
                ok = std::fread(host, sizeof(uint4), n, f) == n &&
                     lbad::hip_status(hipMemcpy(c->d_planes + (size_t)p * c->capacity + at, host, n * sizeof(uint4),
                                                hipMemcpyHostToDevice), "corpus plane H2D", __LINE__) == noErr;
            }
        }
        std::free(host);
        if (!ok) {
            LBAudioDetectiveCorpusDispose(c);
            c = NULL;
        }
    }
    if (c) c->count = h.count;
    std::fclose(f);
    return c;
}

OSStatus LBAudioDetectiveCorpusQuery(LBAudioDetectiveCorpusRef c, LBAudioDetectiveFingerprintRef inQuery, UInt32 inRange,
                                     SInt64* outIndex, Float32* outScore) {
    if (!c) return kLBAudioDetectiveArgumentInvalid;
    unsigned long long key = 0;
    if (inQuery && c->count > 0 && c->variant != 1 && inQuery->length == c->subfp_len &&
        lbad::planes_fast_supported(c->subfp_len, c->n_sub, inQuery->count)) {
        c->appended = false;                                // every append made the query stream wait for it
        OSStatus st = lbad::query_fast(c, inQuery, inRange, &key);
        if (st != noErr) return st;
        LBAudioDetectiveCorpusDecodeKey(key, outIndex, outScore);
        return noErr;
    }
    OSStatus st = lbad::run_query(c, inQuery, inRange, 0, nullptr, c->d_key, nullptr);
    if (st != noErr) return st;
    LBAD_HIP(hipMemcpy(&key, c->d_key, sizeof(key), hipMemcpyDeviceToHost));
    LBAudioDetectiveCorpusDecodeKey(key, outIndex, outScore);
    return noErr;
}

OSStatus LBAudioDetectiveCorpusSetBoundPruning(LBAudioDetectiveCorpusRef c, UInt32 inEnabled) {
    if (!c) return kLBAudioDetectiveArgumentInvalid;
    c->bound_pruning = inEnabled != 0;
    return noErr;
}

OSStatus LBAudioDetectiveCorpusSetBoundPruningThreshold(LBAudioDetectiveCorpusRef c, Float32 inScore) {
    if (!c || !(inScore > 0.0f) || !(inScore <= 1.0f)) return kLBAudioDetectiveArgumentInvalid;
    c->prune_from = inScore;
    return noErr;
}

Float32 LBAudioDetectiveCorpusGetBoundPruningThreshold(LBAudioDetectiveCorpusRef c) { return c ? c->prune_from : 0.0f; }

}  // extern "C"
