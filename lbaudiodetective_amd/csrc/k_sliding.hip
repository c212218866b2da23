// k_sliding.hip -- corpus scan for entries of ANY length: the sliding compare of
// LBAudioDetectiveFingerprintCompareToFingerprint (LBAudioDetectiveFingerprint.m:119-149) with every entry of a
// ragged corpus as the second argument and the query as the first, i.e. the best-match loop of
// LBAudioDetectiveTests/LBAudioDetectiveTests.m:57-91 (one original against sequences of other lengths) as ONE
// launch.
//
// The corpus is a stream of sub-fingerprint RECORDS, 32 bytes each, entries back to back, plus one record offset per
// entry (d_off).  A record (round 4 layout, eight words):
//   w0..w2  P   first Boolean of pairs 0..95  (pairs = ceil(length / 2) <= 100), pair p at bit p & 31 of word p >> 5
//   w3      bits 0..3: P of pairs 96..99;  bits 4..16: the row of the quotient table for this record's `possible`
//           over the FULL range, possible (possible + 1) / 2;  bits 17..20: entry index, bits 28..31;
//           bits 21..24: index of the sub-fingerprint inside its entry, saturated at 15;  bits 25..28: sub-fingerprints
//           that follow it inside its entry, saturated at 15
//   w4..w6  N   second Boolean of pairs 0..95
//   w7      bits 0..3: N of pairs 96..99;  bits 4..31: entry index, bits 0..27
// 25 of the 32 bytes are the reference's information (25 B per sub-fingerprint).  Everything else is
// DERIVED (the table row from the Booleans, the place fields from the entries' counts) and is written by this
// library only: the loader recomputes it from the counts it has validated (restamp_records_kernel), so a corpus file
// cannot plant it.  The bits above the pairs never score: a hit needs both Booleans of a pair equal on both sides,
// and the query's words are zero there.  The place fields serve the scan of SHORT queries only (compare_short_kernel).
//
// Per pair of sub-fingerprints (A = the longer fingerprint's, B = the other's; Fp.m:151-176):
//   NZ = (PA | NA) & RANGE,  possible = popc(NZ),  hits = popc(NZ & ~((PA ^ PB) | (NA ^ NB)))
//   ratio = hits / possible (0 when possible == 0) -- taken from a triangular table of the 5151 correctly rounded
//   quotients in LDS instead of an IEEE division per pair.
//
// Round 4: only the sliding offsets that EXIST are evaluated.  (Round 3 let an accumulator per diagonal travel
// across the lanes of a 256-record chunk: every (query sub-fingerprint, record) pair was computed although 45 % of
// them at a query of 21 -- 76 % at a query of 48 -- lie on diagonals that leave their entry, and chunks overlapped.)
//
//   task  = four consecutive sliding offsets o0 .. o0 + 3 of ONE entry; a lane owns a task and keeps its four
//           float32 sums in place.  The query's sub-fingerprints i = 0, 1, ... are wave-uniform (scalar loads): in
//           step i the lane adds ratio(i, offset) to each of its sums -- the reference's order (Fp.m:139-142).
//   "A"   entry LONGER than the query (Fp.m:123-131 swaps: it becomes fingerprint1): offset o pairs query i with record
//           o + i.  The lane's window of records slides one record per step; the record it needs next is the one
//           its RIGHT neighbour (offsets o0 + 4 ..) used four steps earlier, so records travel leftwards through the
//           lanes of an entry (8 `v_mov_b32 wave_shl:1` per step for 4 pairs) and only the entry's last task reads
//           memory (its fresh record lands five steps before it is needed, directly in the window's ring of 8
//           register slots -- no staging copy).
//   "B"   entry not longer than the query (no swap: the query is fingerprint1): offset o pairs query j with record
//           j - o; records travel rightwards, the entry's FIRST task reads memory, and positions outside the entry
//           are fed from an all-zero record, whose pairs score hits = 0 -> +0.0 -- adding them leaves the float32
//           sum as it is, so the step loop is the same n_query steps for every lane.
//   A wave turns entries into tasks itself: it claims runs of entries from a counter, keeps (first task, records,
//   length) of the entries that still have tasks in a small LDS queue and runs a PASS -- 64 tasks, n_query steps --
//   whenever the queue holds 64 of them; what is left over waits for the next claim, so every pass but a wave's
//   last is full.  All A passes of the scan first, then the B passes (skipped when the host's length histogram
//   says there are none).  No chunk overlap, no diagonal leaves its entry, no side table besides d_off.
//
// max over offsets commutes with the division by n2 (a correctly rounded division by a positive constant is
// monotonic), so there is ONE division per task, and only where the sum can still reach the wave's best (Fp.m:144).
//
// Round 5: the windows of an "A" pass are fetched as whole lines and dealt out through LDS (fill_windows: lane by lane the
// eight loads touched 64 lines each -- a quarter of the scan); feeder loads stop at the last record a pass uses; a pass
// never spans more than 2^26 records (32-bit byte offsets inside it); up to FOUR queries of one length share a pass
// (QN: LBAudioDetectiveTests.m:57-91 is Q originals against N candidates), eight in the systolic scan of short queries;
// a single query travels as a kernel argument and the scan clears its own result words (ScanOut: no copy, no memset on
// the stream in front of a scan).
//
// Round 6: BATCHES of queries of up to 12 sub-fingerprints against the entries longer than them go through a kernel of their
// own, compare_short_multi_kernel (four records per lane, query words in vector registers on register banks the record words
// do not use, pairs 96..99 from an LDS table, chunks claimed from an LDS cursor: eight queries of 5 in 0.63 ms where the
// systolic scan took 1.00); the task scan's query quads lie rotated by one word in LDS for the same bank reason, and no
// instance of either kernel spills a register (tests/test_isa.py checks the compiled code for both).
#include "internal.hpp"

#include <cmath>
#include <type_traits>
#include <utility>
#include <cstring>
#include <mutex>

namespace lbad {
namespace {

constexpr int kSlThreads = 256;       // pack / plan kernels
#ifndef LBAD_SCAN_THREADS
#define LBAD_SCAN_THREADS 1024
#define LBAD_SCAN_PER_CU 1
#endif
constexpr int kScanThreads = LBAD_SCAN_THREADS;    // the scan: ONE workgroup fills a CU (16 waves, four per SIMD)
constexpr int kScanPerCu = LBAD_SCAN_PER_CU;
constexpr int kScanWaves = kScanThreads / 64;
constexpr uint32_t kTriPairs = 100;
constexpr uint32_t kTriSize = (kTriPairs + 1) * (kTriPairs + 2) / 2;   // 5151 quotients
constexpr uint32_t kQWords = 16;      // per query sub-fingerprint: P[4] N[4] NZ[4] tri-base possible - -
constexpr uint32_t kQHeader = 16;     // words in front of the query (reserved, zero)
constexpr uint32_t kSlots = 128;      // queue slots per wave (at most 63 left over + 64 new)
#ifndef LBAD_CLAIM_DIV
#define LBAD_CLAIM_DIV 2u          // a claim takes 1 / (LBAD_CLAIM_DIV * waves) of what is left of the workgroup's run ...
#define LBAD_CLAIM_PASSES 2u       // ... but at least this many passes' worth of entries
#define LBAD_CLAIM_MIN 8u
#endif
// Bound pruning of top-1 scans.  A pass is given up once, for every lane, T = fl(top + n) < stop_below = fl(fl(bs nq) kPruneMargin),
// top = the lane's largest sum so far, n = steps still to come, bs = the best score published.  Why nothing that could win
// or tie is lost: with u = 2^-24, n more float additions of terms <= 1 end at S <= (top + n) (1 + u)^n <= T (1 + (n + 2) u),
// and an entry matters only if fl(S / nq) >= bs, i.e. S >= bs nq (1 - u).  T < bs nq kPruneMargin (1 + 2 u) therefore
// drops nothing as long as kPruneMargin (1 + 2 u) (1 + (n + 2) u) <= 1 - u, which 0.999 satisfies for (n + 5) u <= 0.001:
// queries of up to 16 000 sub-fingerprints (the launcher switches the pruning off beyond kPruneMaxQuery).
#ifndef LBAD_SLIDE_QROT
#define LBAD_SLIDE_QROT 1          // query quads rotated by one word in LDS: no three-sources-on-one-bank v_bitop3 (0: as round 5)
#endif
constexpr uint32_t kPassSpan = 1u << 26;      // records an "A" pass may span: 32-bit byte offsets inside it stay below 2^31
constexpr float kPruneMargin = 0.999f;
constexpr uint32_t kPruneMaxQuery = 8192;

__device__ __forceinline__ unsigned long long sl_key(float score, uint64_t global_index) {
    return ((unsigned long long)__float_as_uint(score) << 32) |
           (unsigned long long)(0xFFFFFFFFu - (uint32_t)global_index);
}

// value of lane l + 1 / lane l - 1 (the wave's last / first lane keeps `old`); full EXEC wherever these are used:
// a DPP read of a lane that is switched off does not deliver its register
__device__ __forceinline__ uint32_t from_right_lane(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t from_left_lane(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xF, 0xF, true);
}

struct SlideArgs {                // the scalars of a scan (the pointers are kernel arguments of their own: `restrict`)
    uint64_t index_base;
    uint64_t n_entries;
    uint32_t nq;
    uint32_t zero_rec;            // index of an all-zero record
    uint32_t rm[4];               // RANGE over pair bits
    uint32_t dense_a, dense_b;    // entries that hold about 128 tasks of a kind (two passes): the least a wave claims
    uint32_t prune;               // 1: a top-1 scan (no per-entry scores wanted) may drop passes that cannot reach the best match so far
    float prune_from;             // ... once a match of at least this score has been published (CorpusSetBoundPruningThreshold)
    uint32_t q_in_args;           // 1: the query block travels in the kernel's argument segment (QueryArg), not through d_query
    uint32_t b_min;               // "B" tasks only for entries of at least this many sub-fingerprints (the shorter ones: systolic scan)
};

// A query of up to kSlideQueryArgSubs sub-fingerprints fits the kernel's argument segment (4 KB): no copy to the device, one
// node less on the stream in front of every scan.
struct QueryArg {
    uint32_t w[(kSlideQueryArgSubs + 1) * 16];
};

// Where a scan leaves its result.  acc: n_keys words that are ZERO between scans (the scan's running maxima, also what a
// strong match is published through while the scan runs); ticket: workgroups that have finished; the last one moves acc to
// keys and clears both -- no memset in front of the scan.
struct ScanOut {
    unsigned long long* acc;
    unsigned int* ticket;
    unsigned long long* keys;
    uint32_t pos[8];              // query i's key goes to keys[pos[i]]
};

struct SlidePtrs {
    const uint4* __restrict__ recs;
    const uint32_t* __restrict__ off;      // n_entries + 1 record positions
    const uint32_t* __restrict__ q;        // the query, kQWords per sub-fingerprint
    unsigned int* score_bits;              // optional, per entry
    unsigned long long* key_out;           // the scan's result word: also where a strong match is published while the scan runs
};

#if defined(LBAD_SLIDE_PROF) || defined(LBAD_SLIDE_STAMPS)
// per wave: start, cursor dry (A), end of A, end (100 MHz stamps); tasks queued when dry, passes after it, their ticks, passes
__device__ unsigned long long g_slide_times[1024 * 16 * 8];
#define LBAD_STAMP(slot, value) do { if ((threadIdx.x & 63u) == 0) g_slide_times[(blockIdx.x * kScanWaves + (threadIdx.x >> 6)) * 8 + (slot)] = (value); } while (0)
#else
#define LBAD_STAMP(slot, value)
#endif
#ifdef LBAD_SLIDE_PROF
// bring-up aid: shader-clock ticks per phase, summed over the waves (tools/exp/sliding_prof.py reads them)
__device__ unsigned long long g_slide_prof[16];
__device__ unsigned int g_slide_passes[256 * 16 * 64 * 4];          // per wave and A pass (up to 64): start (100 MHz, low word), shader ticks       // per wave: start, cursor dry (A), end of A, end -- 100 MHz ticks
#define LBAD_PROF_T(x) const unsigned long long x = __builtin_readcyclecounter()
__shared__ unsigned long long s_slide_prof[kScanWaves][16];     // per wave, flushed once when the kernel ends
#define LBAD_PROF_ADD(slot, a, b) do { if ((threadIdx.x & 63u) == 0) s_slide_prof[threadIdx.x >> 6][slot] += (b) - (a); } while (0)
#else
#define LBAD_PROF_T(x)
#define LBAD_PROF_ADD(slot, a, b)
#endif

struct Task {                     // one lane's share of a pass
    bool active, feeder;
    uint32_t rec0;                // A: record index of offset o0, step 0;  B: first record of the entry
    uint32_t ne, o0, n_off, ent;
};

template <bool FULL>
__device__ __forceinline__ uint32_t record_row(const uint32_t (&r)[8], const uint32_t (&rm)[4]) {
    if (FULL) return (r[3] >> 4) & 0x1FFFu;
    uint32_t p = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) p += __popc((r[w] | r[4 + w]) & rm[w]);   // w3's row bits lie outside rm
    return p * (p + 1u) / 2u;
}

// hits of the window record against the step's query sub-fingerprint, accumulated onto the table row:
//   A: u = (P | N) & ~(P ^ qP)  (0xA4 over P, N, qP; the row bits of w3 meet N = 0, qP = 0 and vanish),
//      v = u & ~(N ^ qN)        (0x90)
//   B: u = nzq & ~(P ^ qP), v = u & ~(N ^ qN): the mask is the query's (it is the longer side), the row bits meet
//      nzq = 0
template <bool MODE_B, bool FULL>
__device__ __forceinline__ uint32_t pair_index(const uint32_t (&r)[8], const uint32_t (&qv)[8], const uint32_t (&nzq)[4],
                                               uint32_t row, const uint32_t (&rm)[4]) {
    uint32_t h = 0;
#ifdef LBAD_EXP_SLIDE_WORDS
    constexpr int kWords = LBAD_EXP_SLIDE_WORDS;
#else
    constexpr int kWords = 4;
#endif
#pragma unroll
    for (int w = 0; w < kWords; ++w) {
        uint32_t u;
        if (MODE_B) u = __builtin_amdgcn_bitop3_b32(nzq[w], r[w], qv[w], 0x90);
        else u = __builtin_amdgcn_bitop3_b32(r[w], r[4 + w], qv[w], 0xA4);
        uint32_t v = __builtin_amdgcn_bitop3_b32(u, r[4 + w], qv[4 + w], 0x90);
        if (!MODE_B && !FULL) v &= rm[w];
        // h += popc(v) as ONE accumulating v_bcnt, starting at the table row
        if (w == 0) asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(h) : "v"(v), "v"(row));
        else asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(h) : "v"(v));
    }
    return h;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// A record slot of a lane's window: words 0..3 and 4..7 as two register quads (the loads' destinations)
struct Slot {
    u32x4 lo, hi;
};

__device__ __forceinline__ void slot_words(const Slot& s, uint32_t (&r)[8]) {
    r[0] = s.lo.x; r[1] = s.lo.y; r[2] = s.lo.z; r[3] = s.lo.w; r[4] = s.hi.x; r[5] = s.hi.y; r[6] = s.hi.z; r[7] = s.hi.w;
}

__device__ __forceinline__ void slot_load(const uint4* __restrict__ recs, uint32_t idx, Slot& s) {
    const uint4* p = recs + 2 * (uint64_t)idx;
    const uint4 a = p[0], b = p[1];
    s.lo = u32x4{a.x, a.y, a.z, a.w};
    s.hi = u32x4{b.x, b.y, b.z, b.w};
}

// The step loop's loads: the lanes of `mask` fetch their next record straight into the slot, the other lanes keep what
// the slot holds.  Written as assembly because the count of loads in flight must not depend on the path taken: the
// compiler skips an exec-masked load when no lane wants it, can then no longer tell how many loads follow a given
// one, and drains the queue (vmcnt(0)) at every loop head -- the five steps a record is fetched ahead would be lost.
// Exactly two loads per step are issued here, whatever the mask, and slot_wait() below counts on that.
__device__ __forceinline__ void slot_load_masked(const uint4* __restrict__ recs, uint32_t idx, unsigned long long mask, Slot& s) {
    const uint4* p = recs + 2 * (uint64_t)idx;
    unsigned long long saved;
    asm volatile("s_mov_b64 %2, exec\n\t"
                 "s_mov_b64 exec, %4\n\t"
                 "global_load_dwordx4 %0, %3, off\n\t"
                 "global_load_dwordx4 %1, %3, off offset:16\n\t"
                 "s_mov_b64 exec, %2"
                 : "+v"(s.lo), "+v"(s.hi), "=&s"(saved)
                 : "v"(p), "s"(mask)
                 : "memory");
}

// The same with the address split into a wave-uniform base (scalar registers: advancing it per step costs no vector
// instruction) and a per-lane byte offset that stays put for the whole pass ("A" passes: a lane's records are consecutive).
__device__ __forceinline__ void slot_load_masked_at(const uint4* base_in, uint32_t byte_off, unsigned long long mask, Slot& s) {
    // (the base IS uniform; saying so keeps it in scalar registers whatever the compiler's divergence analysis concluded)
    const unsigned long long bits = reinterpret_cast<unsigned long long>(base_in);
    const unsigned long long base = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bits) |
                                    ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(bits >> 32)) << 32);
    unsigned long long saved;
    asm volatile("s_mov_b64 %2, exec\n\t"
                 "s_mov_b64 exec, %5\n\t"
                 "global_load_dwordx4 %0, %3, %4\n\t"
                 "global_load_dwordx4 %1, %3, %4 offset:16\n\t"
                 "s_mov_b64 exec, %2"
                 : "+v"(s.lo), "+v"(s.hi), "=&s"(saved)
                 : "v"(byte_off), "s"(base), "s"(mask)
                 : "memory");
}

__device__ __forceinline__ void slot_from_right(Slot& d, const Slot& s) {
    d.lo.x = from_right_lane(s.lo.x); d.lo.y = from_right_lane(s.lo.y); d.lo.z = from_right_lane(s.lo.z); d.lo.w = from_right_lane(s.lo.w);
    d.hi.x = from_right_lane(s.hi.x); d.hi.y = from_right_lane(s.hi.y); d.hi.z = from_right_lane(s.hi.z); d.hi.w = from_right_lane(s.hi.w);
}
__device__ __forceinline__ void slot_from_left(Slot& d, const Slot& s) {
    d.lo.x = from_left_lane(s.lo.x); d.lo.y = from_left_lane(s.lo.y); d.lo.z = from_left_lane(s.lo.z); d.lo.w = from_left_lane(s.lo.w);
    d.hi.x = from_left_lane(s.hi.x); d.hi.y = from_left_lane(s.hi.y); d.hi.z = from_left_lane(s.hi.z); d.hi.w = from_left_lane(s.hi.w);
}

// the loads of the steps after the awaited one may still be in flight (2 per step, IN_FLIGHT of them), everything
// older has landed: memory reads return in order
// f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{}): a step's ring slots are registers, i.e.
// compile-time constants
template <int N, typename F, int... Is>
__device__ __forceinline__ void for_each_slot_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void for_each_slot(F&& f) {
    for_each_slot_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

template <int IN_FLIGHT>
__device__ __forceinline__ void slot_wait_n(Slot& s) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(s.lo), "+v"(s.hi) : "n"(IN_FLIGHT));
}

// The windows of an "A" pass at step 0: four consecutive records = 128 contiguous bytes per lane, and the windows of an
// entry's lanes lie back to back.  Fetched lane by lane (round 4) that is eight load instructions which touch 64 lines
// each, sixteen bytes at a time: the vector L1 serves one line per clock and keeps few of them until the next of the
// eight instructions asks again -- a quarter of the scan's time (knock-out: 0.473 -> 0.356 ms without the fill, 0.400
// with the same bytes fetched as whole lines).  Here the pass's 8 KB are fetched as whole lines and dealt out through LDS:
//   chunk x = 8 t + m (m = 0..7) is half m & 1 of record m >> 1 of task t's window; load j (0..7) of lane l fetches
//   chunk 64 j + l, i.e. bytes 16 (l & 7) .. of task 8 j + (l >> 3) -- eight lanes cover a window's 128 bytes;
//   a half of the tasks at a time (4.5 KB of LDS per wave): four loads are written to LDS, 144 bytes per task (the 16
//   bytes of padding put the sixteen lanes of a ds_read_b128 on distinct banks), and the half's 32 lanes read their
//   eight chunks back.  The first records of the tasks travel the same way (one word each).
constexpr uint32_t kStageTask = 9;                    // uint4 per task in the staging block
constexpr uint32_t kStageWords = 32 * kStageTask + 16;  // uint4 per wave: half a pass + the 64 first-record indices
// `touched`: a register that loads issued before the call are still on their way into (the feeders' line touches of
// run_pass); it stays reserved until they have landed.
__device__ __forceinline__ void fill_windows(const uint4* __restrict__ recs, const Task& t, uint32_t rec_first, uint4* s_stage,
                                             Slot& w0, Slot& w1, Slot& w2, Slot& w3, uint32_t& touched) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t* s_first = reinterpret_cast<uint32_t*>(s_stage + 32 * kStageTask);
    s_first[(lane & 7u) * 8u + (lane >> 3)] = t.active ? t.rec0 : rec_first;     // idle lanes: any record that exists
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint4 f0 = reinterpret_cast<const uint4*>(s_first)[2u * (lane >> 3)];      // first records of tasks (lane >> 3) + 8 j
    const uint4 f1 = reinterpret_cast<const uint4*>(s_first)[2u * (lane >> 3) + 1u];
    const uint32_t first[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
    // all eight loads first, by hand: left to itself the compiler fetches the second half only after the first half has
    // been dealt out (two round trips to memory per pass), and -- seeing no dependence between lanes -- moves a half's
    // LDS writes into the branch of the lanes that read them.  The wave barriers below are what tells it that all 64
    // lanes write before any lane reads.
    u32x4 c[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint4* src = recs + 2 * (uint64_t)first[j] + (lane & 7u);
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(c[j]) : "v"(src) : "memory");
    }
    Slot* w[4] = {&w0, &w1, &w2, &w3};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        // loads return in order: with at most four in flight the first four have landed
        if (h == 0) asm volatile("s_waitcnt vmcnt(4)" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(touched) : : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]) : : "memory");
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const u32x4 v = c[4 * h + jj];
            s_stage[(64u * jj + lane) + (8u * jj + (lane >> 3))] = make_uint4(v.x, v.y, v.z, v.w);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if ((lane >> 5) == (uint32_t)h) {
            const uint4* mine = s_stage + kStageTask * (lane & 31u);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const uint4 lo = mine[2 * n], hi = mine[2 * n + 1];
                w[n]->lo = u32x4{lo.x, lo.y, lo.z, lo.w};
                w[n]->hi = u32x4{hi.x, hi.y, hi.z, hi.w};
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

#ifndef LBAD_SLIDE_RING
#define LBAD_SLIDE_RING 6
#endif
constexpr int kRing = LBAD_SLIDE_RING;        // record slots per lane: the 4 of the window + kRing - 4 fetched ahead
#ifndef LBAD_SLIDE_RING_MULTI
#define LBAD_SLIDE_RING_MULTI 5
#endif
constexpr int kRingMulti = LBAD_SLIDE_RING_MULTI;   // the same for passes of several queries (their steps are longer)

// One pass: 64 tasks, nq steps.  R is the lane's ring of S = kRing record slots.
//   A: slot n % S holds the record of relative index n (records o0 + n of the entry); step i uses n = i + k for the
//      offsets k = 0..3; after the step, record i + S replaces record i: from the right neighbour's record i + S - 4
//      (all lanes, full EXEC), then, in feeder lanes, from memory.
//   B: slot n % S holds e[n - o0]; step j uses n = j - k; after the step record j + S - 3 replaces record j - 3: from
//      the left neighbour's record j + S - 7, then, in feeder lanes, from memory (the zero record outside the entry).
// A record fetched behind step i is first needed S - 4 steps later (A: the left neighbour copies it, one step before
// the lane itself uses it as offset 3; B likewise), and that is where its wait stands.
// Every step fetches, needed or not (an A feeder reads up to S records past its entry: the corpus is allocated
// with that slack), so that the number of loads in flight is the same on every path.
// The step's query sub-fingerprint comes from LDS (QLDS: two ds_read_b128 per step, the same address in every lane)
// rather than from scalar registers: a vector instruction with a scalar operand issues at the slow rate on this chip
// (v_bitop3_b32 1.85 ns against 1.13 ns per SIMD, tools/ubench/slide_rates.hip), and the step has 32 of them.
// QN queries of ONE length in a pass (round 5): the records travel, are fetched and have their table row extracted once;
// a step compares the four window records with sub-fingerprint i of every query (LDS: query qi at s_q + qi (nq + 1)
// kQWords) and a lane keeps 4 QN sums.  QN = 1 is the single scan with its one-step-late table adds; for QN > 1 the
// other queries' arithmetic covers the look-ups.
template <bool MODE_B, bool FULL, bool ALL_FEED, bool QLDS, int QN>
// stop_below (A passes of a top-1 scan, else 0): once every lane's best sum so far plus one whole point per step still to
// come stays under it, nothing in this pass can reach the best match known -- the pass ends and returns false (its sums
// are then meaningless).
__device__ __forceinline__ bool run_pass(const SlideArgs& a, const uint4* __restrict__ recs, const uint32_t* __restrict__ q,
                                         const uint32_t* s_q, const Task& t, const float* s_tri, float (&acc)[QN][4],
                                         const float stop_below, uint4* s_stage) {
    static_assert(QN == 1 || QLDS, "several queries in a pass are read from LDS");
    constexpr int S = QN > 1 ? kRingMulti : kRing;
    constexpr int D = S - 4;
    static_assert(S >= 5 && S <= 8, "ring of 5..8 slots");
    Slot R[S];
    uint32_t row[S];                            // A: table row of the record in slot s (extracted when it becomes offset 3's)
#pragma unroll
    for (int n = 0; n < S; ++n) row[n] = 0;
    const uint32_t nq = a.nq;
    const uint32_t rm[4] = {a.rm[0], a.rm[1], a.rm[2], a.rm[3]};
    const unsigned long long feeders = __ballot(t.feeder);
    // index of relative record n for this lane (B: the zero record outside the entry, and in idle lanes)
    auto rec_index = [&](int32_t n) -> uint32_t {
        if (!MODE_B) return t.rec0 + (uint32_t)n;
        const int32_t p = n - (int32_t)t.o0;
        return (t.active && p >= 0 && (uint32_t)p < t.ne) ? t.rec0 + (uint32_t)p : a.zero_rec;
    };
    auto mod = [](int n) { return ((n % S) + S) % S; };
    // "A" passes: record rec0 + n of a lane lies at a_base + 2 n (uint4 units), a_off bytes further on.  The lanes of a pass
    // are in entry order, lane 0 first; what separates two lanes are whole entries that hold at most 64 tasks together, i.e.
    // fewer than 64 (n_query + 260) records: the offset fits 32 bits with room to spare (the launcher refuses queries of a
    // million sub-fingerprints and more).
    const uint32_t rec_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)t.rec0);
    const uint4* a_base = recs + 2 * (uint64_t)rec_first;
    const uint32_t a_off = MODE_B ? 0u : (t.rec0 - rec_first) * 32u;
    LBAD_PROF_T(q0);
    // The ring at step 0, ONE exposed round trip to memory: every lane fetches the four records of its window; the D
    // records behind it come from the neighbour's window once that has landed, and in feeder lanes from memory -- as
    // masked loads of the step loop's kind, issued in slot order right here, so that the loop's own waits (everything
    // but the last 2 (D - 1) loads has landed) cover them: they are still in flight when step 0 starts.
    if (!MODE_B) {
#if defined(LBAD_EXP_SLIDE_LANEFILL)
        // (round 4's fill: every lane fetches its own 128 bytes -- eight instructions that touch 64 lines each)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            asm volatile("" : "=v"(R[n].lo), "=v"(R[n].hi));       // idle lanes: whatever the registers hold (never used)
            if (t.active) slot_load(recs, rec_index(n), R[n]);
        }
#else
        uint32_t touched = 0;          // (no loads of this pass are in flight yet)
        fill_windows(recs, t, rec_first, s_stage, R[0], R[1], R[2], R[3], touched);
#endif
#pragma unroll
        for (int n = 4; n < S; ++n) {
            if (ALL_FEED) asm volatile("" : "=v"(R[n].lo), "=v"(R[n].hi));
            else slot_from_right(R[n], R[n - 4]);
        }
#pragma unroll
        for (int n = 4; n < S; ++n) slot_load_masked_at(a_base + 2 * n, a_off, feeders, R[n]);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            uint32_t w[8];
            slot_words(R[k], w);
            row[k] = record_row<FULL>(w, rm);
        }
    } else {
        // records -3..0: the zero record unless the entry starts here; 1..D from the left neighbour's -3.. / memory
#pragma unroll
        for (int n = -3; n <= 0; ++n) slot_load(recs, rec_index(n), R[mod(n)]);
#pragma unroll
        for (int n = 1; n <= D; ++n) {
            if (ALL_FEED) asm volatile("" : "=v"(R[mod(n)].lo), "=v"(R[mod(n)].hi));
            else slot_from_left(R[mod(n)], R[mod(n - 4)]);
        }
#pragma unroll
        for (int n = 1; n <= D; ++n) slot_load_masked(recs, rec_index(n), feeders, R[mod(n)]);
    }
    float pend[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // (QN = 1) the previous step's quotients: added one step late, their LDS latency hidden
#pragma unroll
    for (int qi = 0; qi < QN; ++qi)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[qi][k] = 0.0f;
    // every slot is "used" here, in front of the loop: the compiler places the waits for the loads above HERE and not
    // at their first use inside the loop (where a wait would drain the step loop's own loads every S steps)
#pragma unroll
    for (int n = 0; n < S; ++n) asm volatile("" : "+v"(R[n].lo), "+v"(R[n].hi));
    static_assert(D >= 1, "the masked loads of the fill need a slot behind the window");

    // the query sub-fingerprint of a step: P[4] N[4] (and, B: the query's mask and table row) -- fetched one step ahead
    struct QStep {
        uint32_t v[8], nz[4], row;
    };
    const uint32_t q_stride = (nq + 1u) * kQWords;            // words from a query to the next one
    auto fetch_q = [&](uint32_t i, QStep& o, int qi = 0) {
        if (QLDS) {
            const uint32_t* sq = s_q + (size_t)qi * q_stride + (size_t)i * kQWords;
            const uint4* ql = reinterpret_cast<const uint4*>(sq);
#if LBAD_SLIDE_QROT
            // The P and N quads lie in LDS rotated by one word (the kernel's copy loop): word w sits in register base +
            // (w + 1 & 3) of an even-aligned quad, on a bank of the other parity than the record words P[w], N[w] it
            // meets (register base' + w, base' even) -- no v_bitop3 of the step reads three registers of one bank and
            // falls back to the 4-cycle issue (tools/ubench/operand_rates.hip).  The empty asm keeps each read whole.
            const u32x4 q0 = reinterpret_cast<const u32x4*>(sq)[0], q1 = reinterpret_cast<const u32x4*>(sq)[1];
            asm volatile("" :: "v"(q0), "v"(q1));
            o.v[0] = q0.y; o.v[1] = q0.z; o.v[2] = q0.w; o.v[3] = q0.x; o.v[4] = q1.y; o.v[5] = q1.z; o.v[6] = q1.w; o.v[7] = q1.x;
#else
            const uint4 q0 = ql[0], q1 = ql[1];
            o.v[0] = q0.x; o.v[1] = q0.y; o.v[2] = q0.z; o.v[3] = q0.w; o.v[4] = q1.x; o.v[5] = q1.y; o.v[6] = q1.z; o.v[7] = q1.w;
#endif
            if (MODE_B) {
                const uint4 q2 = ql[2];
                o.nz[0] = q2.x; o.nz[1] = q2.y; o.nz[2] = q2.z; o.nz[3] = q2.w;
                o.row = sq[12];
            }
        } else {
            const uint32_t* __restrict__ qa = q + (size_t)i * kQWords;
#pragma unroll
            for (int w = 0; w < 8; ++w) o.v[w] = qa[w];
            if (MODE_B) {
                // the query's mask and table row in vector registers, once per step for the four offsets (a VALU
                // instruction reads one scalar operand only)
#pragma unroll
                for (int w = 0; w < 4; ++w) asm("v_mov_b32 %0, %1" : "=v"(o.nz[w]) : "s"(qa[8 + w]));
                asm("v_mov_b32 %0, %1" : "=v"(o.row) : "s"(qa[12]));
            }
        }
    };
#ifdef LBAD_SLIDE_PROF
    // (profiling build) wait for the fill here so that its duration can be read off
#pragma unroll
    for (int n = 0; n < S; ++n) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(R[n].lo), "+v"(R[n].hi) : : "memory");
    LBAD_PROF_T(q1);
    LBAD_PROF_ADD(4, q0, q1);
    if ((threadIdx.x & 63u) == 0) s_slide_prof[threadIdx.x >> 6][15] = q1 - q0;
#endif
    QStep qn;
#pragma unroll
    for (int w = 0; w < 4; ++w) qn.nz[w] = 0;
    qn.row = 0;
    if (QN == 1) fetch_q(0, qn);
    // one step; u = i % S is a compile-time constant (the ring's slots are registers)
    auto step = [&](auto u_c, const uint32_t i) {
        constexpr int u = decltype(u_c)::value;
        if (!MODE_B) {
            uint32_t w[8];
            slot_words(R[(u + 3) % S], w);
            row[(u + 3) % S] = record_row<FULL>(w, rm);
        }
        if constexpr (QN == 1) {
            const QStep qc = qn;
#ifndef LBAD_EXP_SLIDE_NOQ
            fetch_q(i + 1u, qn);                                // (one sub-fingerprint of slack behind the query)
#endif
            uint32_t h[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t w[8];
                slot_words(R[MODE_B ? mod(u - k) : (u + k) % S], w);
                h[k] = pair_index<MODE_B, FULL>(w, qc.v, qc.nz, MODE_B ? qc.row : row[(u + k) % S], rm);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc[0][k] = __fadd_rn(acc[0][k], pend[k]);      // step i - 1's term (0.0 in front of the first)
#ifdef LBAD_EXP_SLIDE_NOTRI
                pend[k] = __uint_as_float(h[k]);
#else
                pend[k] = s_tri[h[k]];
#endif
            }
        } else {
#pragma unroll
            for (int qi = 0; qi < QN; ++qi) {
                QStep qc;
#pragma unroll
                for (int w = 0; w < 4; ++w) qc.nz[w] = 0;
                qc.row = 0;
                fetch_q(i, qc, qi);
                uint32_t h[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    uint32_t w[8];
                    slot_words(R[MODE_B ? mod(u - k) : (u + k) % S], w);
                    h[k] = pair_index<MODE_B, FULL>(w, qc.v, qc.nz, MODE_B ? qc.row : row[(u + k) % S], rm);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[qi][k] = __fadd_rn(acc[qi][k], s_tri[h[k]]);    // in step order (Fp.m:139-142)
            }
        }
        // the window moves on: the record fetched D steps ago has landed by now
#ifdef LBAD_SLIDE_PROF_STEP
        LBAD_PROF_T(w0);
#endif
        if (!MODE_B) {
#ifndef LBAD_EXP_SLIDE_NOLOADS
            slot_wait_n<2 * (D - 1)>(R[(u + 4) % S]);
#endif
#ifdef LBAD_SLIDE_PROF_STEP
            LBAD_PROF_T(w1);
            LBAD_PROF_ADD(13, w0, w1);
            LBAD_PROF_ADD(14, 0ull, 1ull);
#endif
#ifndef LBAD_EXP_SLIDE_NODPP
            if (!ALL_FEED) slot_from_right(R[u], R[(u + D) % S]);
#endif
#if defined(LBAD_EXP_SLIDE_NOFEED)
            slot_load_masked_at(a_base + 2 * (size_t)(i + (uint32_t)S), a_off, 0ull, R[u]);
#elif !defined(LBAD_EXP_SLIDE_NOLOADS)
            // the last record any offset of the pass uses is relative record nq + 2 (offset 3, step nq - 1): the loads
            // behind it still issue (the waits count instructions) but with no lane switched on -- round 4 read up to
            // six records past every entry's end, 13 % of the scan's traffic
            unsigned long long fnow;                            // (scalar by hand: the compiler selects 64-bit values in vector registers)
            asm("s_cmp_le_u32 %1, %2\n\ts_cselect_b64 %0, %3, 0" : "=s"(fnow) : "s"((uint32_t)__builtin_amdgcn_readfirstlane((int)(i + (uint32_t)S))), "s"(nq + 2u), "s"(feeders) : "scc");
            slot_load_masked_at(a_base + 2 * (size_t)(i + (uint32_t)S), a_off, fnow, R[u]);
#endif
        } else {
            slot_wait_n<2 * (D - 1)>(R[mod(u + 1)]);
            if (!ALL_FEED) slot_from_left(R[mod(u - 3)], R[mod(u + S - 7)]);
            slot_load_masked(recs, rec_index((int32_t)i + S - 3), feeders, R[mod(u - 3)]);
        }
    };
    // whole rounds of the ring without a branch inside (one block for the scheduler: the next step's query words and
    // this step's table look-ups are in flight across the steps), then the last nq % S steps one by one
    uint32_t i0 = 0;
    bool alive = true;
    for (; i0 + (uint32_t)S <= nq; i0 += (uint32_t)S) {
        for_each_slot<S>([&](auto u_c) { step(u_c, i0 + (uint32_t)decltype(u_c)::value); });
        if (QN == 1 && !MODE_B && stop_below > 0.0f) {          // (uniform) a ratio is at most 1: an upper bound of every sum
            const float top = fmaxf(fmaxf(acc[0][0] + pend[0], acc[0][1] + pend[1]), fmaxf(acc[0][2] + pend[2], acc[0][3] + pend[3]));
            if (!__any(t.active && top + (float)(nq - (i0 + (uint32_t)S)) >= stop_below)) { alive = false; break; }
        }
    }
    if (alive) {
        for_each_slot<S - 1>([&](auto u_c) {
            if (i0 + (uint32_t)decltype(u_c)::value < nq) step(u_c, i0 + (uint32_t)decltype(u_c)::value);   // uniform
        });
    }
    if (QN == 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[0][k] = __fadd_rn(acc[0][k], pend[k]);
    }
    // Nothing of this pass may still be in flight when the ring's registers are reused.  The wait NAMES every slot:
    // the compiler does not know that loads are on their way into them, sees the ring dead after the last step and
    // would otherwise hand its registers to the code that follows (scheduled in front of a wait without operands) --
    // the late records then land in the task's results.
#pragma unroll
    for (int n = 0; n < S; ++n) asm volatile("s_waitcnt vmcnt(0)" : "+v"(R[n].lo), "+v"(R[n].hi) : : "memory");
    return alive;
}

__device__ __forceinline__ uint32_t wave_inclusive_add(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)v, d, 64);
        if (lane >= (uint32_t)d) v += t;
    }
    return v;
}

// One wave's part of the A (or B) tasks of its workgroup; returns the best key it met.
//
// Who does what.  A workgroup is a whole CU (16 waves) and owns a run of entries that holds 1 / grid of the scan's tasks
// (the plan below: made once per query length).  Inside the workgroup the waves CLAIM pieces of that run from a cursor
// in LDS -- 64 entries while the run is long, down to 8 near its end -- so all sixteen stay busy until the run is
// empty.  Two other arrangements were measured on the 1 M-entry corpus first: claims from ONE counter in global memory
// (16 k same-address atomics per scan: a fifth of a wave's time went into waiting for them, plus the tail of whoever
// claimed last) and fully static shares per wave (no atomics at all, but waves that share a SIMD do not run at one
// speed: the slowest needed 1.37 x the average and the rest of its SIMD idled meanwhile).
template <bool MODE_B, bool FULL, bool ALL_FEED, bool QLDS, int QN, int WAVES>
__device__ __forceinline__ void scan_mode(const SlideArgs& a, const SlidePtrs& p, const float* s_tri, const uint32_t* s_q,
                                          uint32_t* s_cursor, uint32_t* s_start, uint32_t* s_off, uint32_t* s_ne,
                                          uint32_t* s_ent, uint4* s_stage, unsigned long long (&best)[QN]) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t nq = a.nq;
    const uint32_t run_end = s_cursor[1];
    uint32_t n_slots = 0, done = 0, total = 0;       // queue: slots, tasks handed out, tasks queued (this wave's numbering)
    uint32_t cur = 0, cur_end = 0;                   // claimed entries not yet turned into slots
    bool more = true;
    if (lane == 0) s_start[0] = 0;
    for (;;) {
        LBAD_PROF_T(p0);
        // ---- refill: until 64 tasks wait or the workgroup's run is exhausted -----------------------------------
        while (more && total - done < 64u) {
            if (cur >= cur_end) {
                uint32_t c0 = 0, size = 0;
                if (lane == 0) {
                    // a 32nd of what is left, but never less than two passes' worth of entries (where the tasks of a
                    // kind are rare -- the few entries not longer than a short query -- sixteen waves with a handful of
                    // tasks each would run sixteen nearly empty passes) nor fewer than 8, and at most 4096 entries
                    const uint32_t seen = *reinterpret_cast<volatile uint32_t*>(s_cursor);
                    const uint32_t left = seen < run_end ? run_end - seen : 0u;
                    const uint32_t least = MODE_B ? a.dense_b : a.dense_a;
                    size = left / (LBAD_CLAIM_DIV * (uint32_t)WAVES);
                    size = size < least ? least : size;
                    size = size < LBAD_CLAIM_MIN ? LBAD_CLAIM_MIN : (size > 4096u ? 4096u : size);
                    c0 = atomicAdd(s_cursor, size);
                }
                c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)c0);
                size = (uint32_t)__builtin_amdgcn_readfirstlane((int)size);
                if (c0 >= run_end) {
                    if (!MODE_B) {
                        LBAD_STAMP(1, __builtin_amdgcn_s_memrealtime());
                        LBAD_STAMP(4, total - done);
                    }
                    more = false;
                    break;
                }
                cur = c0;
                cur_end = run_end - c0 < size ? run_end : c0 + size;
            }
            // drop the slots whose tasks have all been handed out (a prefix: the starts are increasing)
            {
                const bool live0 = lane < n_slots && s_start[lane + 1] > done;
                const bool live1 = lane + 64u < n_slots && s_start[lane + 65u] > done;
                const uint32_t n_live = (uint32_t)__popcll(__ballot(live0)) + (uint32_t)__popcll(__ballot(live1));
                const uint32_t first = n_slots - n_live;
                if (first) {
                    uint32_t v[2][4];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t src = first + lane + 64u * h;
                        const bool ok = src < n_slots;
                        v[h][0] = ok ? s_start[src] : 0u; v[h][1] = ok ? s_off[src] : 0u;
                        v[h][2] = ok ? s_ne[src] : 0u; v[h][3] = ok ? s_ent[src] : 0u;
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t dst = lane + 64u * h;
                        if (first + dst < n_slots) {
                            s_start[dst] = v[h][0]; s_off[dst] = v[h][1]; s_ne[dst] = v[h][2]; s_ent[dst] = v[h][3];
                        }
                    }
                    n_slots = n_live;
                    if (lane == 0) s_start[n_slots] = total;
                }
            }
            // the next (at most 64) claimed entries
            const uint32_t e = cur + lane;
            const uint32_t part_end = cur_end - cur > 64u ? cur + 64u : cur_end;
            uint32_t o = 0, ne = 0, tasks = 0;
            if (e < part_end) {
                o = p.off[e];
                ne = p.off[(uint64_t)e + 1] - o;
                // Fp.m:123-131 swaps only when the query is SHORTER: an entry of the query's length is a "B" entry
                if (MODE_B) tasks = (ne <= nq && ne >= a.b_min) ? (nq - ne + 4u) >> 2 : 0u;   // ceil((nq - ne + 1) / 4)
                else tasks = ne > nq ? (ne - nq + 4u) >> 2 : 0u;
            }
            cur = part_end;
            const unsigned long long with = __ballot(tasks != 0u);
            const uint32_t incl = wave_inclusive_add(tasks, lane);
            if (tasks) {
                const uint32_t slot = n_slots + (uint32_t)__popcll(with & ((1ull << lane) - 1ull));
                s_start[slot] = total + incl - tasks;
                s_off[slot] = o; s_ne[slot] = ne; s_ent[slot] = e;
            }
            n_slots += (uint32_t)__popcll(with);
            total += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (lane == 0) s_start[n_slots] = total;
        }
        if (total == done) break;
        LBAD_PROF_T(p1);
        LBAD_PROF_ADD(0, p0, p1);
        // ---- one pass over the next 64 tasks ------------------------------------------------------------------
        Task t;
        const uint32_t tid = done + lane;
#ifdef LBAD_EXP_SLIDE_PARTIAL
        // (experiment) every pass takes at most LBAD_EXP_SLIDE_PARTIAL tasks: are partly filled passes slow by themselves?
        uint32_t pass_tasks = total - done < LBAD_EXP_SLIDE_PARTIAL ? total - done : LBAD_EXP_SLIDE_PARTIAL;
#else
        uint32_t pass_tasks = total - done < 64u ? total - done : 64u;
#endif
        t.active = lane < pass_tasks;
        const uint32_t key_t = t.active ? tid : done + pass_tasks - 1u;
        uint32_t lo = 0, hi = n_slots;                       // s_start[lo] <= key_t < s_start[hi]
#pragma unroll
        for (int it = 0; it < 7; ++it) {
            const uint32_t mid = (lo + hi) >> 1;
            const bool right = s_start[mid] <= key_t;
            lo = right ? mid : lo;
            hi = right ? hi : mid;
        }
        const uint32_t tl = key_t - s_start[lo];             // task inside the entry
        t.ne = s_ne[lo];
        t.ent = s_ent[lo];
        t.o0 = 4u * tl;
        t.n_off = MODE_B ? nq - t.ne + 1u : t.ne - nq + 1u;
        const uint32_t e_first = s_off[lo];
        t.rec0 = MODE_B ? e_first : e_first + t.o0;
        if (!MODE_B) {
            // run_pass addresses a lane's records as a wave-uniform base + a 32-bit byte offset.  The lanes of a pass are in
            // entry order, but entries WITHOUT tasks of this kind (and entries other waves claimed) may lie between two of
            // them -- in a corpus of more than 4 GiB a wave's left-over tasks and its next claim can be 2^27 records apart
            // (round-4 advice).  A pass therefore takes only the leading tasks within kPassSpan records of its first one;
            // the others stay queued and open the next pass.
            const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)t.rec0);
            const uint32_t near = (uint32_t)__popcll(__ballot(t.active && t.rec0 - first < kPassSpan));
            if (near < pass_tasks) {                         // (uniform; a prefix: the records ascend with the lanes)
                pass_tasks = near;
                t.active = lane < pass_tasks;
            }
        }
        // who reads memory: the lane whose neighbour cannot hand it the next record -- the entry's last (A) / first (B)
        // task and the pass's edge lanes (an entry whose tasks lie in two passes)
        if (ALL_FEED) t.feeder = t.active;
        else if (MODE_B) t.feeder = t.active && (tl == 0u || lane == 0u);
        else t.feeder = t.active && (t.o0 + 4u >= t.n_off || lane + 1u == pass_tasks);
        float acc[QN][4];
#ifdef LBAD_SLIDE_PROBE
        {   // (profiling build) one timed load of a line nobody has touched lately: the memory latency at this moment
            const uint32_t far_rec = (uint32_t)(((uint64_t)t.rec0 * 2654435761ull + 12345ull) % (uint64_t)p.off[a.n_entries]);
            const unsigned long long m0 = __builtin_readcyclecounter();
            const uint32_t pv = *(reinterpret_cast<const volatile uint32_t*>(p.recs + 2 * (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)far_rec)));
            asm volatile("s_waitcnt vmcnt(0)" : : "v"(pv) : "memory");
            const unsigned long long m1 = __builtin_readcyclecounter();
            // ... and of a line that is in the L2 for sure (the offsets of this wave's entries, just read by the refill)
            const uint32_t pw = *(reinterpret_cast<const volatile uint32_t*>(p.off + (uint32_t)__builtin_amdgcn_readfirstlane((int)t.ent)));
            asm volatile("s_waitcnt vmcnt(0)" : : "v"(pw) : "memory");
            const unsigned long long m2 = __builtin_readcyclecounter();
            // ... and a SCALAR load of a line of the query block (scalar cache / L2, not the vector memory path)
            uint32_t sv;
            const uint32_t* sp = p.q + ((uint32_t)__builtin_amdgcn_readfirstlane((int)(t.ent & 15u)) * 16u);
            asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(sv) : "s"(sp) : "memory");
            const unsigned long long m3 = __builtin_readcyclecounter();
            if (lane == 0) s_slide_prof[threadIdx.x >> 6][12] = (m1 - m0) | ((m2 - m1) << 32);
            if (lane == 0) s_slide_prof[threadIdx.x >> 6][11] = (m3 - m2) + (sv == 0x1234567u ? 1u : 0u);
        }
#endif
        LBAD_PROF_T(p2);
        LBAD_PROF_ADD(1, p1, p2);
#ifdef LBAD_SLIDE_PROF
        const unsigned long long r2 = __builtin_amdgcn_s_memrealtime();
#endif
        // Top-1 scans: a strong match (score >= kPruneFrom) found by ANY wave is published in the scan's result word at once;
        // every wave looks at that word before a pass and lets run_pass give up a pass that cannot reach it (an upper bound:
        // exact -- nothing that could win or tie is dropped; per-entry scores are never asked for together with this).
        float stop_below = 0.0f;
        if (QN == 1 && !MODE_B && a.prune) {
            const unsigned long long seen = __hip_atomic_load(p.key_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            best[0] = seen > best[0] ? seen : best[0];
            const float bs = __uint_as_float((uint32_t)(best[0] >> 32));
            if (bs >= a.prune_from) stop_below = bs * (float)nq * kPruneMargin;
        }
        const bool alive = run_pass<MODE_B, FULL, ALL_FEED, QLDS, QN>(a, p.recs, p.q, s_q, t, s_tri, acc, stop_below, s_stage);
        LBAD_PROF_T(p3);
        LBAD_PROF_ADD(2, p2, p3);
#ifdef LBAD_SLIDE_PROF
        if (!MODE_B && lane == 0 && blockIdx.x < 256) {
            const uint32_t wv = blockIdx.x * kScanWaves + (threadIdx.x >> 6);
            const uint32_t ord = (uint32_t)g_slide_times[wv * 8 + 7];
            if (ord < 64) {
                g_slide_passes[(wv * 64 + ord) * 4] = (uint32_t)__builtin_amdgcn_s_memrealtime();
                g_slide_passes[(wv * 64 + ord) * 4 + 1] = (uint32_t)(p3 - p2);
                g_slide_passes[(wv * 64 + ord) * 4 + 2] = (uint32_t)(s_slide_prof[threadIdx.x >> 6][12] >> 32) | ((uint32_t)s_slide_prof[threadIdx.x >> 6][11] << 16);
                g_slide_passes[(wv * 64 + ord) * 4 + 3] = (uint32_t)s_slide_prof[threadIdx.x >> 6][12];
            }
            g_slide_times[wv * 8 + 7] = ord + 1;
        }
        if (!MODE_B && !more && lane == 0) {
            g_slide_times[(blockIdx.x * kScanWaves + (threadIdx.x >> 6)) * 8 + 5] += 1;
            g_slide_times[(blockIdx.x * kScanWaves + (threadIdx.x >> 6)) * 8 + 6] += p3 - p2;
        }
#endif
        LBAD_PROF_ADD(5, 0ull, 1ull);
        done += pass_tasks;
#ifdef LBAD_SLIDE_DEBUG
        if (t.active)
            printf("wg %u wave %u mode %d lane %u ent %u ne %u o0 %u n_off %u rec0 %u feeder %d acc %g %g %g %g total %u n_slots %u\n",
                   blockIdx.x, threadIdx.x >> 6, (int)MODE_B, lane, t.ent, t.ne, t.o0, t.n_off, t.rec0, (int)t.feeder, acc[0][0], acc[0][1],
                   acc[0][2], acc[0][3], total, n_slots);
#endif

        // max over the task's offsets first, ONE exact division where the sum can still matter (Fp.m:144)
        const float n2f = (float)(MODE_B ? t.ne : nq);
#pragma unroll
        for (int qi = 0; qi < QN; ++qi) {
            int m = -1;                                      // sums are >= 0: as integers their bits order like the floats
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (alive && t.active && t.o0 + (uint32_t)k < t.n_off) m = max(m, __float_as_int(acc[qi][k]));
            const float thr = __uint_as_float((uint32_t)(best[qi] >> 32)) * 0.99999f;
            const bool need = m >= 0 && ((QN == 1 && p.score_bits != nullptr) || __int_as_float(m) >= thr * n2f);
            if (need) {
                const float cand = __fdiv_rn(__int_as_float(m), n2f);
                const float match = (0.0f < cand) ? cand : 0.0f;             // MAX(match, cand) from match = 0
                if (QN == 1 && p.score_bits) atomicMax(&p.score_bits[t.ent], __float_as_uint(match));
                const unsigned long long key = sl_key(match, a.index_base + t.ent);
                best[qi] = key > best[qi] ? key : best[qi];
                if (QN == 1 && !MODE_B && a.prune && match >= a.prune_from) atomicMax(p.key_out, key);   // (rare: only a real match gets here)
            }
        }
        LBAD_PROF_T(p4);
        LBAD_PROF_ADD(3, p3, p4);
    }
}

// the query in LDS (dynamic, 64 bytes per sub-fingerprint): up to kQueryLds sub-fingerprints; longer queries are read
// through the scalar cache
constexpr uint32_t kQueryLds = 480;

// starts_a / starts_b: grid + 1 entry indices: workgroup g owns the entries [starts[g], starts[g + 1])
// QN queries of one length per launch (1: the single scan, 1024 threads; 2, 4: 768 threads -- twelve waves leave a lane
// the registers for 4 QN sums); q: QN blocks of (nq + 1) kQWords words (QN = 1: or qa, the same block as an argument).
template <bool FULL, bool ALL_FEED, bool QLDS, int QN, int THREADS>
__global__ __launch_bounds__(THREADS, 1) void compare_sliding_kernel(
    const uint4* __restrict__ recs, const uint32_t* __restrict__ off, const uint32_t* __restrict__ q,
    const uint32_t* __restrict__ starts_a, const uint32_t* __restrict__ starts_b, const float* __restrict__ tri_tbl,
    unsigned int* score_bits, const ScanOut out, const SlideArgs a, const QueryArg qa) {
    constexpr int WAVES = THREADS / 64;
    static_assert(WAVES <= kScanWaves, "the profiling arrays are sized for sixteen waves");
    __shared__ float s_tri[kTriSize];
    __shared__ uint32_t s_queue[WAVES][4 * kSlots + 4];
    __shared__ unsigned long long s_k[WAVES][QN];
    __shared__ uint32_t s_cursor[2][2];                                    // per mode: next entry to claim, end of the run
    __shared__ __attribute__((aligned(16))) uint4 s_stage_all[WAVES][kStageWords];   // fill_windows' staging block, per wave
    extern __shared__ __attribute__((aligned(16))) uint32_t s_qbuf[];      // QLDS: QN * (nq + 1) * kQWords words
    for (uint32_t i = threadIdx.x; i < kTriSize; i += THREADS) s_tri[i] = tri_tbl[i];
    if (QLDS) {
        const uint32_t words = (uint32_t)QN * (a.nq + 1u) * kQWords;
        // (the P and N quads of a sub-fingerprint rotated by one word: see fetch_q)
        auto place = [](uint32_t i) -> uint32_t { return (LBAD_SLIDE_QROT && (i & 15u) < 8u) ? ((i & ~3u) | ((i + 1u) & 3u)) : i; };
        if (QN == 1 && a.q_in_args) {
            for (uint32_t i = threadIdx.x; i < words; i += THREADS) s_qbuf[place(i)] = qa.w[i];
        } else {
            for (uint32_t i = threadIdx.x; i < words; i += THREADS) s_qbuf[place(i)] = q[i];
        }
    }
    if (threadIdx.x == 0) {
        s_cursor[0][0] = starts_a[blockIdx.x]; s_cursor[0][1] = starts_a[blockIdx.x + 1];
        s_cursor[1][0] = starts_b[blockIdx.x]; s_cursor[1][1] = starts_b[blockIdx.x + 1];
    }
    __syncthreads();
    // (the wave's number is wave-uniform: held in a scalar register it costs no vector register across the scan -- as a
    // vector value it was the one spilled register of the single-query instance at the 128-register cap)
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
    uint32_t* s_start = s_queue[wave];            // kSlots + 1 task starts
    uint32_t* s_off = s_start + kSlots + 4;
    uint32_t* s_ne = s_off + kSlots;
    uint32_t* s_ent = s_ne + kSlots;
    SlidePtrs p;
    p.recs = recs; p.off = off; p.q = q; p.score_bits = score_bits; p.key_out = out.acc;
    const uint32_t* s_q = QLDS ? s_qbuf : nullptr;
    unsigned long long best[QN];
#pragma unroll
    for (int qi = 0; qi < QN; ++qi) best[qi] = 0ull;
#ifdef LBAD_SLIDE_PROF
    if (lane < 16) s_slide_prof[wave][lane] = 0ull;
#endif
    LBAD_PROF_T(k0);
#if defined(LBAD_SLIDE_PROF) || defined(LBAD_SLIDE_STAMPS)
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    // Two of the waves take the B entries first, the others the A entries; whoever runs out moves on to the
    // other kind.  Where B entries are rare (a short query: the few entries not longer than it) finding them is a walk
    // over the whole run's offsets -- latency, not arithmetic -- and hides behind the others' A passes this way.
    const bool b_first = wave >= (uint32_t)WAVES - 2u;
    if (b_first) scan_mode<true, FULL, ALL_FEED, QLDS, QN, WAVES>(a, p, s_tri, s_q, s_cursor[1], s_start, s_off, s_ne, s_ent, s_stage_all[wave], best);
    scan_mode<false, FULL, ALL_FEED, QLDS, QN, WAVES>(a, p, s_tri, s_q, s_cursor[0], s_start, s_off, s_ne, s_ent, s_stage_all[wave], best);
    LBAD_PROF_T(k1);
    LBAD_STAMP(0, rt0);
    LBAD_STAMP(2, __builtin_amdgcn_s_memrealtime());
    if (!b_first) scan_mode<true, FULL, ALL_FEED, QLDS, QN, WAVES>(a, p, s_tri, s_q, s_cursor[1], s_start, s_off, s_ne, s_ent, s_stage_all[wave], best);
    LBAD_PROF_T(k2);
    LBAD_PROF_ADD(6, k0, k1);
    LBAD_PROF_ADD(7, k1, k2);
    LBAD_PROF_ADD(8, 0ull, 1ull);
#ifdef LBAD_SLIDE_PROF
    LBAD_PROF_ADD(9, rt0, __builtin_amdgcn_s_memrealtime());      // 100 MHz ticks of the wave's whole scan
    if (lane == 0) {
        const unsigned long long busy = __builtin_amdgcn_s_memrealtime() - rt0;
        atomicMax(&g_slide_prof[10], busy);
        atomicMin(&g_slide_prof[11], busy);
    }
    if (lane < 10 || lane == 13 || lane == 14) atomicAdd(&g_slide_prof[lane], s_slide_prof[wave][lane]);
#endif
    LBAD_STAMP(3, __builtin_amdgcn_s_memrealtime());
#pragma unroll
    for (int qi = 0; qi < QN; ++qi) {
#pragma unroll
        for (int off2 = 32; off2 > 0; off2 >>= 1) {
            const unsigned long long o = __shfl_xor(best[qi], off2, 64);
            best[qi] = o > best[qi] ? o : best[qi];
        }
        if (lane == 0) s_k[wave][qi] = best[qi];
    }
    __syncthreads();
    // The workgroup's maxima join the scan's running ones; the LAST workgroup to get here hands the results over and
    // leaves the running words and the ticket at zero for the next scan (no memset node in front of a scan).
    if (threadIdx.x == 0) {
        for (int qi = 0; qi < QN; ++qi) {
            unsigned long long m = s_k[0][qi];
            for (int i = 1; i < WAVES; ++i) m = s_k[i][qi] > m ? s_k[i][qi] : m;
            if (m) atomicMax(&out.acc[qi], m);
        }
        __threadfence();
        if (atomicAdd(out.ticket, 1u) == gridDim.x - 1u) {
            __threadfence();
            for (int qi = 0; qi < QN; ++qi) out.keys[out.pos[qi]] = atomicExch(&out.acc[qi], 0ull);
            *out.ticket = 0u;
            __threadfence();
        }
    }
}

// ---- the plan of a query length: where every workgroup's run of entries starts ---------------------------------------
// tasks of an entry of ne sub-fingerprints against a query of nq: groups of four of its |ne - nq| + 1 sliding offsets;
// Fp.m:123-131 swaps only when the query is SHORTER, so an entry of the query's length slides along the query ("B")
// (b_min: "B" entries shorter than this are not the task kernel's -- launch_compare_sliding hands them to the systolic scan)
__device__ __forceinline__ void entry_tasks(uint32_t ne, uint32_t nq, uint32_t b_min, uint32_t& ta, uint32_t& tb) {
    ta = ne > nq ? (ne - nq + 4u) >> 2 : 0u;
    tb = (ne <= nq && ne >= b_min) ? (nq - ne + 4u) >> 2 : 0u;
}
constexpr uint32_t kPlanPerThread = 4, kPlanPerBlock = kSlThreads * kPlanPerThread;

// sums of a block of 1024 entries -> block_sums[2 b], [2 b + 1]
__global__ __launch_bounds__(kSlThreads) void slide_plan_sums_kernel(const uint32_t* __restrict__ off, uint64_t n_entries, uint32_t nq,
                                                                    uint32_t b_min, uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t s_a[kSlThreads / 64], s_b[kSlThreads / 64];
    const uint64_t e0 = (uint64_t)blockIdx.x * kPlanPerBlock + (uint64_t)threadIdx.x * kPlanPerThread;
    uint32_t sa = 0, sb = 0;
#pragma unroll
    for (uint32_t k = 0; k < kPlanPerThread; ++k) {
        const uint64_t e = e0 + k;
        if (e < n_entries) {
            uint32_t ta, tb;
            entry_tasks(off[e + 1] - off[e], nq, b_min, ta, tb);
            sa += ta; sb += tb;
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { sa += __shfl_xor((int)sa, d, 64); sb += __shfl_xor((int)sb, d, 64); }
    if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = sa; s_b[threadIdx.x >> 6] = sb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t ta = 0, tb = 0;
        for (int i = 0; i < kSlThreads / 64; ++i) { ta += s_a[i]; tb += s_b[i]; }
        block_sums[2 * blockIdx.x] = ta;
        block_sums[2 * blockIdx.x + 1] = tb;
    }
}

// exclusive scan of the block sums in place (one workgroup); presets the run starts: starts[0] = 0, the others =
// n_entries (a workgroup whose share begins behind the last task owns nothing)
__global__ __launch_bounds__(1024) void slide_plan_scan_kernel(uint32_t* __restrict__ block_sums, uint32_t n_blocks, uint32_t n_entries,
                                                                uint32_t grid, uint32_t chunk_a, uint32_t chunk_b,
                                                                uint32_t* __restrict__ starts_a, uint32_t* __restrict__ starts_b) {
    __shared__ uint32_t s_wa[16], s_wb[16];
    __shared__ uint32_t s_carry[2];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    for (uint32_t g = threadIdx.x; g <= grid; g += 1024u) {
        starts_a[g] = (g && chunk_a) ? n_entries : 0u;      // no tasks of a kind: every run of that kind is empty
        starts_b[g] = (g && chunk_b) ? n_entries : 0u;
    }
    if (threadIdx.x == 0) { s_carry[0] = 0; s_carry[1] = 0; }
    __syncthreads();
    for (uint32_t base = 0; base < n_blocks; base += 1024u) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t va = i < n_blocks ? block_sums[2 * i] : 0u, vb = i < n_blocks ? block_sums[2 * i + 1] : 0u;
        uint32_t ia = va, ib = vb;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t ta = (uint32_t)__shfl_up((int)ia, d, 64), tb = (uint32_t)__shfl_up((int)ib, d, 64);
            if (lane >= (uint32_t)d) { ia += ta; ib += tb; }
        }
        if (lane == 63) { s_wa[wv] = ia; s_wb[wv] = ib; }
        __syncthreads();
        uint32_t oa = s_carry[0], ob = s_carry[1];
        for (uint32_t k = 0; k < wv; ++k) { oa += s_wa[k]; ob += s_wb[k]; }
        if (i < n_blocks) {
            block_sums[2 * i] = oa + ia - va;
            block_sums[2 * i + 1] = ob + ib - vb;
        }
        __syncthreads();
        if (threadIdx.x == 1023) { s_carry[0] = oa + ia; s_carry[1] = ob + ib; }
        __syncthreads();
    }
}

// starts[g] = the first entry whose tasks begin at or behind task g * chunk (tasks in front of it >= g * chunk): a
// workgroup owns WHOLE entries, its share differs from chunk by less than one entry's tasks
__global__ __launch_bounds__(kSlThreads) void slide_plan_final_kernel(const uint32_t* __restrict__ off, uint64_t n_entries, uint32_t nq,
                                                                     uint32_t b_min, const uint32_t* __restrict__ block_sums, uint32_t chunk_a,
                                                                     uint32_t chunk_b, uint32_t* __restrict__ starts_a,
                                                                     uint32_t* __restrict__ starts_b) {
    __shared__ uint32_t s_a[kSlThreads / 64], s_b[kSlThreads / 64];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint64_t e0 = (uint64_t)blockIdx.x * kPlanPerBlock + (uint64_t)threadIdx.x * kPlanPerThread;
    uint32_t ta[kPlanPerThread], tb[kPlanPerThread], sa = 0, sb = 0;
#pragma unroll
    for (uint32_t k = 0; k < kPlanPerThread; ++k) {
        ta[k] = tb[k] = 0;
        if (e0 + k < n_entries) entry_tasks(off[e0 + k + 1] - off[e0 + k], nq, b_min, ta[k], tb[k]);
        sa += ta[k]; sb += tb[k];
    }
    uint32_t ia = sa, ib = sb;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t xa = (uint32_t)__shfl_up((int)ia, d, 64), xb = (uint32_t)__shfl_up((int)ib, d, 64);
        if (lane >= (uint32_t)d) { ia += xa; ib += xb; }
    }
    if (lane == 63) { s_a[wv] = ia; s_b[wv] = ib; }
    __syncthreads();
    uint64_t pa = (uint64_t)block_sums[2 * blockIdx.x] + ia - sa, pb = (uint64_t)block_sums[2 * blockIdx.x + 1] + ib - sb;
    for (uint32_t k = 0; k < wv; ++k) { pa += s_a[k]; pb += s_b[k]; }
#pragma unroll
    for (uint32_t k = 0; k < kPlanPerThread; ++k) {
        const uint64_t e = e0 + k;
        if (e < n_entries) {
            // entry e + 1 is the first with >= g chunk tasks in front of it for every g with pa < g chunk <= pa + ta
            if (ta[k] && chunk_a)
                for (uint64_t g = pa / chunk_a + 1u; g * chunk_a <= pa + ta[k]; ++g) starts_a[g] = (uint32_t)(e + 1u);
            if (tb[k] && chunk_b)
                for (uint64_t g = pb / chunk_b + 1u; g * chunk_b <= pb + tb[k]; ++g) starts_b[g] = (uint32_t)(e + 1u);
            pa += ta[k]; pb += tb[k];
        }
    }
}

// ---- short queries (up to 7 sub-fingerprints): the round-3 systolic scan --------------------------------------------
// A lane holds ONE record (two aligned, fully coalesced dwordx4 per lane: every record is read once, 16 cache lines per
// wave instruction -- the task kernel above reads a lane's four-record window from 64 different lines and is bound by
// the texture path when the steps are few), the query's sub-fingerprints a = 0, 1, ... are wave-uniform, and an
// accumulator per sliding offset travels one lane to the right per step (`v_add_f32 ... wave_shr:1`).  Every (query,
// record) pair is evaluated, also on the diagonals that leave their entry -- with a query of q against entries of n
// that is (q - 1) / n of the work, a tenth at q = 5 -- and the scan is HBM-bound.  Chunks of 64 records overlap by
// min(n_query, longest entry) - 1.  The place of a record inside its entry comes from the record (w3 / w7, see the top of
// the file): no side table, no search.
//   entry longer than the query ("A" lanes): a diagonal starts in step 0 in every lane and is complete after the last
//     step; it is an offset of the entry iff it stayed inside the entry (i >= n_query - 1).
//   entry not longer than the query ("B" lanes): a diagonal starts whenever it enters the entry's first lane (i == 0) and
//     is complete when it leaves the last one (r == 0); that lane keeps the maximum over the steps.
// Diagonals that did not start properly carry -inf.
struct Rec {
    uint32_t P[4], N[4];
    uint32_t isat, rem, idx;
};

__device__ __forceinline__ Rec unpack_rec(const uint4 a, const uint4 b) {
    Rec r;
    r.P[0] = a.x; r.P[1] = a.y; r.P[2] = a.z; r.P[3] = a.w & 0xFu;
    r.N[0] = b.x; r.N[1] = b.y; r.N[2] = b.z; r.N[3] = b.w & 0xFu;
    r.isat = (a.w >> 21) & 0xFu;
    r.rem = (a.w >> 25) & 0xFu;
    r.idx = (b.w >> 4) | ((a.w >> 17) & 0xFu) << 28;
    return r;
}

constexpr int kNegInf = (int)0xFF800000u;   // -inf; as a signed integer it sorts below the bits of every sum >= 0

// One chunk = 64 K consecutive records; lane l holds records K l .. K l + K - 1, so a diagonal moves from register
// set k - 1 to set k inside the lane and crosses to the next lane only from set K - 1 to set 0.
// MODE 0: every entry in the chunk is longer than the query; 1: none is; 2: mixed.
// QN queries of one length (round 5): step a of ALL queries before step a + 1 -- their sums travel side by side, a query's
// chain of dependent adds and look-ups is covered by the others' bit operations (query qi at q + qi q_stride).
template <int K, int MODE, int QN>
__device__ __forceinline__ void short_steps(const uint32_t (&P)[K][4], const uint32_t (&N)[K][4], const uint32_t (&nz)[K][4],
                                          const uint32_t (&tri)[K], const bool (&case_a)[K], const bool (&start_b)[K],
                                          const uint32_t* __restrict__ q, uint32_t q_stride, uint32_t nq, const float* s_tri,
                                          float (&acc)[QN][K], int (&smax)[QN][K]) {
    // MODE 2: the mask of a cell is the entry's NZ in A lanes and the query's in B lanes:
    //   m = sel_e & (nz_q | sel_a)   with sel_e = A ? nz_e : ~0,  sel_a = A ? ~0 : 0     (one v_bitop3)
    uint32_t sel_e[K][4], sel_a[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
        for (int qi = 0; qi < QN; ++qi) {
            acc[qi][k] = MODE == 0 ? 0.0f : __int_as_float(kNegInf);   // MODE 0: step 0 adds to the zeros, no reset needed
            smax[qi][k] = kNegInf;
        }
        sel_a[k] = case_a[k] ? 0xFFFFFFFFu : 0u;
#pragma unroll
        for (int w = 0; w < 4; ++w) sel_e[k][w] = case_a[k] ? nz[k][w] : 0xFFFFFFFFu;
    }
    for (uint32_t a = 0; a < nq; ++a) {
#pragma unroll
        for (int qi = 0; qi < QN; ++qi) {
            const uint32_t* __restrict__ qa = q + (size_t)qi * q_stride + (size_t)a * kQWords;
            uint32_t nzq[4] = {0, 0, 0, 0}, triq = 0;
            if (MODE != 0) {
                // the query's mask and table row in vector registers, once per step for the K cells (a VALU
                // instruction reads one scalar operand only)
#pragma unroll
                for (int w = 0; w < 4; ++w) asm("v_mov_b32 %0, %1" : "=v"(nzq[w]) : "s"(qa[8 + w]));
                asm("v_mov_b32 %0, %1" : "=v"(triq) : "s"(qa[12]));
            }
            float ratio[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint32_t h0 = MODE == 0 ? tri[k] : MODE == 1 ? triq : (case_a[k] ? tri[k] : triq);
                uint32_t h = 0;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    // a & ~(b ^ c), twice: two v_bitop3_b32 per word (left to itself the compiler builds xor, xor, bitop3)
                    uint32_t m = MODE == 0 ? nz[k][w] : nzq[w];
                    if (MODE == 2) m = __builtin_amdgcn_bitop3_b32(sel_e[k][w], nzq[w], sel_a[k], 0xE0);   // a & (b | c)
                    const uint32_t u = __builtin_amdgcn_bitop3_b32(m, P[k][w], qa[w], 0x90);
                    const uint32_t v = __builtin_amdgcn_bitop3_b32(u, N[k][w], qa[4 + w], 0x90);
                    // h += popc(v) as ONE accumulating v_bcnt (the compiler distributes the table's * 4 over the sum)
                    if (w == 0) asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(h) : "v"(v), "v"(h0));   // starts at the table row
                    else asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(h) : "v"(v));
                }
                ratio[k] = s_tri[h];
            }
            const float in0 = __uint_as_float(from_left_lane(__float_as_uint(acc[qi][K - 1])));
#pragma unroll
            for (int k = K - 1; k >= 0; --k) {
                float sh = k ? acc[qi][k - 1] : in0;
                if (MODE == 1) sh = start_b[k] ? 0.0f : sh;
                if (MODE == 2) sh = (start_b[k] || (a == 0 && case_a[k])) ? 0.0f : sh;
                acc[qi][k] = __fadd_rn(sh, ratio[k]);
                if (MODE != 0) smax[qi][k] = max(smax[qi][k], __float_as_int(acc[qi][k]));
            }
        }
    }
}

// QN queries of one length per launch (round 5): a chunk's records are fetched and unpacked once, the steps run per query
// (q: QN blocks of (nq + 1) kQWords words).
template <int K, int QN>
__global__ __launch_bounds__(kSlThreads, (K == 4 && QN == 1) ? 4 : 1) void compare_short_kernel(     // (four waves per SIMD: 128 registers, as round 4's)
    const uint4* __restrict__ recs, uint64_t n_pos, const uint32_t* __restrict__ q, uint32_t nq, uint32_t chunk_step,
    uint64_t n_chunks, uint4 range_mask, const float* __restrict__ tri_tbl, uint64_t index_base,
    unsigned int* __restrict__ score_bits, const ScanOut out, uint32_t only_upto) {
    // only_upto (0: every entry): the scan of a corpus that is split between the two kernels -- only entries of at most
    // this many sub-fingerprints are scored here (the task kernel has the others), chunks without one are passed over
    __shared__ float s_tri[kTriSize];
    __shared__ unsigned long long s_k[kSlThreads / 64][QN];
    for (uint32_t i = threadIdx.x; i < kTriSize; i += kSlThreads) s_tri[i] = tri_tbl[i];
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = (uint64_t)blockIdx.x * (kSlThreads / 64) + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (kSlThreads / 64);
    const uint32_t rm[4] = {range_mask.x, range_mask.y, range_mask.z, range_mask.w};
    unsigned long long best[QN];
#pragma unroll
    for (int qi = 0; qi < QN; ++qi) best[qi] = 0ull;
    const uint32_t q_stride = (nq + 1u) * kQWords;

    for (uint64_t c = wave; c < n_chunks; c += n_waves) {
        const uint64_t p0 = c * chunk_step + (uint64_t)lane * K;
        uint32_t P[K][4], N[K][4], nz[K][4], tri[K], isat[K], rem[K], idx[K], n2[K];
        bool case_a[K], start_b[K], inb[K];
        bool some_a = false, some_b = false;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            inb[k] = p0 + k < n_pos;
            uint4 ra = make_uint4(0, 0, 0, 0), rb = make_uint4(0, 0, 0, 0);
            if (inb[k]) {
                ra = recs[2 * (p0 + k)];
                rb = recs[2 * (p0 + k) + 1];
            }
            const Rec r = unpack_rec(ra, rb);
            uint32_t possible = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                P[k][w] = r.P[w];
                N[k][w] = r.N[w];
                nz[k][w] = (r.P[w] | r.N[w]) & rm[w];
                possible += __popc(nz[k][w]);
            }
            tri[k] = possible * (possible + 1u) / 2u;
            isat[k] = r.isat; rem[k] = r.rem; idx[k] = r.idx;
            const uint32_t ne = r.isat + r.rem + 1u;       // saturated (both fields at 15); exact whenever it is <= 16
            case_a[k] = ne > nq;                           // the entry is the longer side (Fp.m:123-131)
            n2[k] = case_a[k] ? nq : ne;
            start_b[k] = !case_a[k] && r.isat == 0u;
            if (only_upto && ne > only_upto) inb[k] = false;   // (not this scan's entry: its lanes never score)
            some_a |= case_a[k] && inb[k];
            some_b |= !case_a[k] && inb[k];
        }
        const bool any_a = __ballot(some_a) != 0ull, any_b = __ballot(some_b) != 0ull;
        if (only_upto && !any_a && !any_b) continue;       // (uniform)

        float acc[QN][K];
        int smax[QN][K];
        if (!any_b) short_steps<K, 0, QN>(P, N, nz, tri, case_a, start_b, q, q_stride, nq, s_tri, acc, smax);
        else if (!any_a) short_steps<K, 1, QN>(P, N, nz, tri, case_a, start_b, q, q_stride, nq, s_tri, acc, smax);
        else short_steps<K, 2, QN>(P, N, nz, tri, case_a, start_b, q, q_stride, nq, s_tri, acc, smax);
#pragma unroll
        for (int qi = 0; qi < QN; ++qi) {

            // a record closes a window iff the window lies inside its entry AND inside this chunk.  The exact
            // division (Fp.m:144) runs only where the sum can reach the lane's best so far.
            const float thr = __uint_as_float((uint32_t)(best[qi] >> 32)) * 0.99999f;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const bool closes = case_a[k] ? (isat[k] >= nq - 1u) : (rem[k] == 0u);
                const bool valid = inb[k] && closes && lane * K + k >= n2[k] - 1u;
                const float s = case_a[k] ? acc[qi][k] : __int_as_float(smax[qi][k]);
                const float n2f = (float)n2[k];
                const bool need = valid && ((QN == 1 && score_bits != nullptr) || s >= thr * n2f);
                if (__ballot(need) != 0ull) {
                    if (need) {
                        const float cand = __fdiv_rn(s, n2f);
                        const float match = (0.0f < cand) ? cand : 0.0f;     // MAX(match, cand) from match = 0
                        if (QN == 1 && score_bits) atomicMax(&score_bits[idx[k]], __float_as_uint(match));
                        const unsigned long long key = sl_key(match, index_base + idx[k]);
                        best[qi] = key > best[qi] ? key : best[qi];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int qi = 0; qi < QN; ++qi) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(best[qi], off, 64);
            best[qi] = o > best[qi] ? o : best[qi];
        }
        if (lane == 0) s_k[threadIdx.x >> 6][qi] = best[qi];
    }
    __syncthreads();
    // The keys are max-ed in place (the host clears them in front of the launch): this scan runs six workgroups per CU,
    // and a ticket on top of the maximum -- two contended atomics and two fences for each of 1536 workgroups -- cost a
    // fifth of the HBM-bound scan's time (0.261 -> 0.309 ms at a query of 5) where the memset node costs 5 us.
    if (threadIdx.x == 0) {
        for (int qi = 0; qi < QN; ++qi) {
            unsigned long long m = s_k[0][qi];
            for (int i = 1; i < kSlThreads / 64; ++i) m = s_k[i][qi] > m ? s_k[i][qi] : m;
            if (m) atomicMax(&out.keys[out.pos[qi]], m);
        }
    }
}

// ---- round 6: SEVERAL short queries per launch ----------------------------------------------------------------------------
// compare_short_kernel<1, 8> above spent 22 vector instructions per (record, query sub-fingerprint) pair at 4 cycles each:
// every v_bitop3 took a query word as its SCALAR operand, one v_mov_dpp per pair moved the partial sum to the next lane, the
// division branch of the epilogue was taken by every chunk (a lane's own best is beaten by one of its next four windows
// with probability 1 / chunks seen, and one lane of 64 is enough), and the 40 table look-ups per record met on the LDS
// banks (63 % of the LDS cycles).  This kernel keeps the systolic idea (the records rest, the partial sums of a diagonal
// travel) and changes what surrounds it:
//   * FOUR records per lane (a chunk = 256 records per wave): a diagonal crosses a lane boundary once per four pairs;
//   * the query words of step a come from LDS as two broadcast ds_read_b128 per (query, step) into VECTOR registers and
//     serve the lane's four records;
//   * NO v_bitop3 READS THREE REGISTERS OF ONE BANK.  tools/ubench/operand_rates.hip (profiles/r06_operand_rates.txt): a
//     wave64 v_bitop3 / v_fma_f32 issues in 2 cycles unless all three sources lie on one of the four register banks
//     (register number mod 4), then in 4 -- and a compiler that puts every 16-byte load into a 4-aligned quad makes
//     (P[w], N[w], qP[w]) exactly such a triple (25 of the 32 bit operations of a step in the first build).  The record
//     words are used where the loads put them (register base even + w: the tuples of gfx950 are even-aligned) and the
//     query quads are stored ROTATED by one word (word w in register base + (w + 1 & 3)): a query word's bank differs in
//     parity from the bank of the record words it meets, whatever the allocator does;
//   * the records are not unpacked: the query words are cut to the RANGE instead (a pair outside it -- and every field
//     bit of w3 / w7 -- meets query Booleans 0 0 and can only "match" where the record's pair is 0 0 as well, which is no
//     hit); the mask of a pair is folded into the first bit operation (0xA4: (P | N) & ~(P ^ qP));
//   * one query after the other (their sums are independent): four running sums per lane, whatever the number of queries;
//   * only entries LONGER than the query are scored here ("A" lanes: a diagonal starts in step 0 in every lane and is an
//     offset of its entry iff it stayed inside it); the host sends the entries of at most n_query sub-fingerprints, where the
//     corpus has any, through compare_short_kernel in its only_upto mode (same keys, atomic maxima);
//   * RATIO = hits / possible without the table: with pf = (float)possible, rh = RN(1 / pf) and
//     rl = RN(fma(-pf, rh, 1) * rh), fma(hf, rh, RN(hf * rl)) IS the correctly rounded quotient for every
//     0 <= hits <= possible <= 100 (tools/verify_ratio_fma.c checks all 5151 pairs against the IEEE division with the very
//     operations used here); possible == 0 gives rh = rl = 0 -> +0.0 like the table's row 0.  (rh, rl) of the 101 values
//     of `possible` sit in LDS; hits are counted on top of the bits of 2^23, so hf is one subtraction;
//   * the exact division of the epilogue runs where a sum can reach the WAVE's best so far (a wave-uniform threshold,
//     refreshed where the branch is taken: about ln(chunks) times per wave and query instead of every time).
//   * the LAST FOUR PAIRS (96..99: a word of their own in either plane, two bit operations and a count for 4 % of the pairs)
//     come from LDS instead: per (query, step) a table of the 256 tails a record can have (its four P and four N Booleans)
//     holds 2^23's bits + the tail's hits -- the value the three remaining counts start from; the query length is a
//     template argument, so table and query words are read at immediate offsets (14.5 instead of 16.5 vector
//     instructions per pair; the look-ups meet on the banks, but nothing else uses the LDS here);
//   * ONE workgroup of sixteen waves per CU owns a contiguous run of chunks and its waves CLAIM them from a cursor in LDS.
//     With equal static shares the waves did not finish together: the SIMD serves its oldest wave first, the workgroups
//     placed first ended at 0.51 ms, the last at 1.09 (tools/exp/short_multi_stamps.py), and a SIMD's last wave, alone,
//     issues at a fraction of the rate four waves reach together -- the vector ALU idled 60 % of the scan.
#ifndef LBAD_SHORT_MULTI_MAX
#define LBAD_SHORT_MULTI_MAX 12
#endif
// longest query of a BATCH this kernel takes (instantiated for 1..12; the records' place fields reach 15).  Crossover re-measured in round 6,
// eight queries against 1 M entries of 20..70: 1.00 / 1.05 / 1.14 / 1.36 ms at 8 / 9 / 10 / 12 here, 1.24 / 1.25 / 1.34 / 1.45 through the task kernel.
static_assert(LBAD_SHORT_MULTI_MAX >= 7 && LBAD_SHORT_MULTI_MAX <= 12, "compare_short_multi_kernel is instantiated for query lengths 1..12");
constexpr int kShortMultiK = 4;
constexpr int kSmThreads = 1024;

template <int QN, int NQ>
__global__ __launch_bounds__(kSmThreads, 1) void compare_short_multi_kernel(
    const uint4* __restrict__ recs, uint64_t n_pos, const uint32_t* __restrict__ q, uint32_t chunk_step,
    uint64_t n_chunks, uint64_t chunks_per_group, uint4 range_mask, uint64_t index_base, const ScanOut out) {
    constexpr int K = kShortMultiK;
    constexpr uint32_t nq = NQ;
    __shared__ uint4 s_q[QN * NQ * 2];                           // per (query, step): P words, N words, each rotated by one
    __shared__ uint32_t s_tail[QN * NQ][256];                    // per (query, step) and record tail: bits of 2^23 + hits in pairs 96..99
    __shared__ float2 s_rr[kTriPairs + 1];                       // (rh, rl) of possible = 0 .. 100
    __shared__ unsigned long long s_k[kSmThreads / 64][QN];
    __shared__ unsigned int s_cursor;                            // chunks of this workgroup's run handed out so far
    const uint32_t q_stride = (nq + 1u) * kQWords;
    const uint32_t rm[4] = {range_mask.x, range_mask.y, range_mask.z, range_mask.w};
    for (uint32_t i = threadIdx.x; i < QN * nq * 2u; i += kSmThreads) {
        const uint32_t qi = i / (2u * nq), rest = i - qi * 2u * nq;              // rest = 2 a + (0: P, 1: N)
        const uint32_t* src = q + (size_t)qi * q_stride + (size_t)(rest >> 1) * kQWords + (rest & 1u) * 4u;
        s_q[i] = make_uint4(src[3] & rm[3], src[0] & rm[0], src[1] & rm[1], src[2] & rm[2]);
    }
    for (uint32_t i = threadIdx.x; i < QN * nq * 256u; i += kSmThreads) {
        const uint32_t qa = i >> 8, t = i & 255u, qi = qa / nq, a = qa - qi * nq;
        const uint32_t* src = q + (size_t)qi * q_stride + (size_t)a * kQWords;
        const uint32_t qp = src[3] & rm[3], qn = src[7] & rm[3], p = t & 15u, n = t >> 4;
        s_tail[qa][t] = 0x4B000000u + (uint32_t)__popc((p | n) & ~(p ^ qp) & ~(n ^ qn) & 15u);
    }
    if (threadIdx.x == 0) s_cursor = 0u;
    if (threadIdx.x <= kTriPairs) {
        const float pf = (float)threadIdx.x;
        const float r1 = threadIdx.x ? __fdiv_rn(1.0f, pf) : 0.0f;
        s_rr[threadIdx.x] = make_float2(r1, __fmul_rn(__fmaf_rn(-pf, r1, threadIdx.x ? 1.0f : 0.0f), r1));
    }
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t run_first = (uint64_t)blockIdx.x * chunks_per_group;
    const uint64_t run_left = run_first < n_chunks ? n_chunks - run_first : 0ull;
    const uint32_t run_chunks = (uint32_t)(run_left < chunks_per_group ? run_left : chunks_per_group);
    // the next chunk of the run (>= run_chunks: none left); one LDS atomic per wave and chunk
    auto claim = [&]() -> uint32_t {
        uint32_t got = 0;
        if (lane == 0) got = atomicAdd(&s_cursor, 1u);
        return __builtin_amdgcn_readfirstlane(got);
    };
    const float nqf = (float)nq;
    // wave-uniform: the wave's best key so far and what a sum must reach to matter (scalar registers: they change only in
    // the rare division branch, where the lanes' candidates are reduced over the wave at once)
    unsigned long long best[QN];
    float wthr[QN];
#pragma unroll
    for (int qi = 0; qi < QN; ++qi) { best[qi] = 0ull; wthr[qi] = 0.0f; }

    // The records of the NEXT chunk are requested before this chunk's steps run (a second set of 32 registers): waves that
    // run equal phases fall into step -- all of a CU's waves waited for their records at the same time, 0.36 of 0.88 ms with
    // nothing issued (knock-out LBAD_EXP_SM_NOLOAD).
    uint4 na[K], nb[K];
    auto request = [&](uint64_t c) {
        const uint64_t p0 = c * chunk_step + (uint64_t)lane * K;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            na[k] = make_uint4(0, 0, 0, 0);
            nb[k] = make_uint4(0, 0, 0, 0);
#ifdef LBAD_EXP_SM_NOLOAD
            na[k] = make_uint4(lane, k, c, 7);
            nb[k] = make_uint4(k, lane, 5, 0x50 + (lane << 8));
#else
            if (p0 + k < n_pos) {
                na[k] = recs[2 * (p0 + k)];
                nb[k] = recs[2 * (p0 + k) + 1];
            }
#endif
        }
    };
#ifdef LBAD_SLIDE_STAMPS
#define LBAD_SM_STAMP(slot) do { if (lane == 0) g_slide_times[(blockIdx.x * (kSmThreads / 64) + (threadIdx.x >> 6)) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    uint32_t chunks_done = 0;
#else
#define LBAD_SM_STAMP(slot)
#endif
    LBAD_SM_STAMP(0);
    uint32_t cur = claim(), next = run_chunks;
    if (cur < run_chunks) request(run_first + cur);
    for (; cur < run_chunks; cur = next) {
        const uint64_t c = run_first + cur;
#ifdef LBAD_SLIDE_STAMPS
        if (chunks_done == 1) LBAD_SM_STAMP(1);
        if (chunks_done == 17) LBAD_SM_STAMP(2);
        ++chunks_done;
#endif
        const uint64_t p0 = c * chunk_step + (uint64_t)lane * K;
        uint4 ra[K], rb[K];
        float rh[K], rl[K];
        const uint32_t* tail[K];                           // the record's row of s_tail[0]
        bool valid[K];
#pragma unroll
        for (int k = 0; k < K; ++k) { ra[k] = na[k]; rb[k] = nb[k]; }
        next = claim();
        if (next < run_chunks) request(run_first + next);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const Rec r = unpack_rec(ra[k], rb[k]);
            uint32_t possible = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) possible += __popc((r.P[w] | r.N[w]) & rm[w]);
            const float2 rr = s_rr[possible];
            rh[k] = rr.x;
            rl[k] = rr.y;
            tail[k] = &s_tail[0][r.P[3] | r.N[3] << 4];
            const uint32_t ne = r.isat + r.rem + 1u;       // saturated (both fields at 15); exact whenever it is <= 16
            // a record closes a window of an "A" entry iff the window lies inside its entry AND inside this chunk
            valid[k] = p0 + k < n_pos && ne > nq && r.isat >= nq - 1u && lane * K + k >= nq - 1u;
        }

#pragma unroll
        for (int qi = 0; qi < QN; ++qi) {
            float acc[K];
#pragma unroll
            for (int k = 0; k < K; ++k) acc[k] = 0.0f;
#ifdef LBAD_EXP_SM_STEPS
#pragma unroll
            for (int a = 0; a < LBAD_EXP_SM_STEPS; ++a) {
#else
#pragma unroll
            for (int a = 0; a < NQ; ++a) {
#endif
                const u32x4 qp4 = reinterpret_cast<const u32x4*>(s_q)[(qi * NQ + a) * 2];
                const u32x4 qn4 = reinterpret_cast<const u32x4*>(s_q)[(qi * NQ + a) * 2 + 1];
                // (both stay whole 16-byte reads into register quads: as three single words the rotation's parity argument
                // would no longer hold -- the compiler narrows a read whose .x nobody uses)
                asm volatile("" :: "v"(qp4), "v"(qn4));
                const uint32_t qP[3] = {qp4.y, qp4.z, qp4.w}, qN[3] = {qn4.y, qn4.z, qn4.w};    // (rotated by one; .x = pairs 96..99: in the table)
                float ratio[K];
                uint32_t h[K];
#pragma unroll
                for (int k = 0; k < K; ++k) h[k] = tail[k][(qi * NQ + a) * 256];
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const uint32_t P[3] = {ra[k].x, ra[k].y, ra[k].z}, N[3] = {rb[k].x, rb[k].y, rb[k].z};
#pragma unroll
                    for (int w = 0; w < 3; ++w) {
                        const uint32_t u = __builtin_amdgcn_bitop3_b32(P[w], N[w], qP[w], 0xA4);   // (P | N) & ~(P ^ qP)
                        const uint32_t v = __builtin_amdgcn_bitop3_b32(u, N[w], qN[w], 0x90);      // u & ~(N ^ qN)
                        asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(h[k]) : "v"(v));
                    }
                    const float hf = __fsub_rn(__uint_as_float(h[k]), 8388608.0f);
                    const float t = __fmul_rn(hf, rl[k]);
                    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(ratio[k]) : "v"(hf), "v"(rh[k]), "v"(t));
                }
                const float in0 = __uint_as_float(from_left_lane(__float_as_uint(acc[K - 1])));
#pragma unroll
                for (int k = K - 1; k >= 0; --k) {
                    // (as single adds in place: packed, the compiler renames the sums with three moves per step)
                    const float from = k ? acc[k - 1] : in0;
                    asm("v_add_f32 %0, %1, %2" : "=v"(acc[k]) : "v"(from), "v"(ratio[k]));
                }
                // (a step at a time: left alone, the scheduler pulls the LDS reads of every unrolled step to the front and
                // spills 240 registers)
                __builtin_amdgcn_sched_barrier(0);
            }
            // The exact division (Fp.m:144) runs only where the sum can reach the wave's best so far.
            float m = __int_as_float(kNegInf);
#pragma unroll
            for (int k = 0; k < K; ++k) m = fmaxf(m, valid[k] ? acc[k] : __int_as_float(kNegInf));
#ifdef LBAD_EXP_SM_NOEPI
            if (__ballot(m >= 1e30f) != 0ull) {
#else
            if (__ballot(m >= wthr[qi]) != 0ull) {
#endif
                unsigned long long mine = 0ull;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (valid[k] && acc[k] >= wthr[qi]) {
                        const float cand = __fdiv_rn(acc[k], nqf);
                        const float match = (0.0f < cand) ? cand : 0.0f;     // MAX(match, cand) from match = 0
                        const uint32_t idx = (rb[k].w >> 4) | ((ra[k].w >> 17) & 0xFu) << 28;       // (unpack_rec's idx)
                        const unsigned long long key = sl_key(match, index_base + idx);
                        mine = key > mine ? key : mine;
                    }
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const unsigned long long o = __shfl_xor(mine, off, 64);
                    mine = o > mine ? o : mine;
                }
                const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(mine >> 32));
                const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)mine);
                const unsigned long long wave_key = ((unsigned long long)hi << 32) | lo;
                if (wave_key > best[qi]) best[qi] = wave_key;
                // (scores are >= +0: their bits order like the values)
                wthr[qi] = __uint_as_float((uint32_t)(best[qi] >> 32)) * 0.99999f * nqf;
            }
        }
    }
    LBAD_SM_STAMP(3);
    if (lane == 0) {
#pragma unroll
        for (int qi = 0; qi < QN; ++qi) s_k[threadIdx.x >> 6][qi] = best[qi];
    }
    __syncthreads();
    if (threadIdx.x == 0) {                       // the keys are max-ed in place (the host clears them in front of the launch)
        for (int qi = 0; qi < QN; ++qi) {
            unsigned long long m = s_k[0][qi];
            for (int i = 1; i < kSmThreads / 64; ++i) m = s_k[i][qi] > m ? s_k[i][qi] : m;
            if (m) atomicMax(&out.keys[out.pos[qi]], m);
        }
    }
}

// even-position bits of a 32-bit word, compacted into 16
__device__ __forceinline__ uint32_t even_bits(uint32_t x) {
    x &= 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0F0F0F0Fu;
    x = (x | (x >> 4)) & 0x00FF00FFu;
    x = (x | (x >> 8)) & 0x0000FFFFu;
    return x;
}

// the record of the eight words P[0..3], N[0..3] (pairs beyond 99 cleared) with its derived fields: the full-range table
// row and its place (entry `ent`, sub-fingerprint i of it, r more behind it)
__device__ __forceinline__ void store_record(uint4* __restrict__ recs, uint64_t p, const uint32_t (&P)[4], const uint32_t (&N)[4],
                                             uint32_t ent, uint32_t i, uint32_t r) {
    const uint32_t p3 = P[3] & 0xFu, n3 = N[3] & 0xFu;
    const uint32_t possible = __popc(P[0] | N[0]) + __popc(P[1] | N[1]) + __popc(P[2] | N[2]) + __popc(p3 | n3);
    const uint32_t row = possible * (possible + 1u) / 2u;
    const uint32_t isat = i < 15u ? i : 15u, rsat = r < 15u ? r : 15u;
    recs[2 * p] = make_uint4(P[0], P[1], P[2], p3 | (row << 4) | ((ent >> 28) << 17) | (isat << 21) | (rsat << 25));
    recs[2 * p + 1] = make_uint4(N[0], N[1], N[2], n3 | (ent << 4));
}

// entry e with off[e] <= p < off[e + 1] among n entries (off: n + 1 increasing record positions)
__device__ __forceinline__ uint64_t entry_of(const uint32_t* __restrict__ off, uint64_t n, uint64_t p) {
    uint64_t lo = 0, hi = n;
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (off[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
}

// packed sub-fingerprints (8-word slots, Boolean b at bit b) of whole entries -> records.  off: ABSOLUTE record
// positions of the new entries (n_new + 1 values); slot t becomes record off[0] + t
__global__ __launch_bounds__(kSlThreads) void pack_records_kernel(const uint32_t* __restrict__ slots, uint64_t n_new_pos,
                                                                  const uint32_t* __restrict__ off, uint64_t n_new,
                                                                  uint32_t first_entry, uint4* __restrict__ recs) {
    const uint64_t t = (uint64_t)blockIdx.x * kSlThreads + threadIdx.x;
    if (t >= n_new_pos) return;
    const uint64_t p = (uint64_t)off[0] + t;
    const uint64_t e = entry_of(off, n_new, p);
    const uint4* s = reinterpret_cast<const uint4*>(slots + t * kPackedWords);
    const uint4 a = s[0], b = s[1];
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    // pairs 0..99 live in bits 0..199 = words 0..6 (word 6: 8 bits)
    uint32_t P[4], N[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t w0 = w[2 * k], w1 = 2 * k + 1 < 7 ? w[2 * k + 1] : 0u;
        P[k] = even_bits(w0) | (even_bits(w1) << 16);
        N[k] = even_bits(w0 >> 1) | (even_bits(w1 >> 1) << 16);
    }
    store_record(recs, p, P, N, first_entry + (uint32_t)e, (uint32_t)p - off[e], off[e + 1] - 1u - (uint32_t)p);
}

// Records that come from a FILE: keep the 200 Booleans, recompute everything derived (table row; place fields from the
// offsets the loader built out of the validated counts), clear everything reserved.
// old_layout: the round-3 file ("LBADCRP2": P at bits 0..99, N at bits 100..199, place fields above).
__global__ __launch_bounds__(kSlThreads) void restamp_records_kernel(uint4* __restrict__ recs, const uint32_t* __restrict__ off,
                                                                     uint64_t n_entries, uint64_t n, uint32_t old_layout,
                                                                     uint4 pair_mask) {
    const uint64_t t = (uint64_t)blockIdx.x * kSlThreads + threadIdx.x;
    if (t >= n) return;
    const uint4 a = recs[2 * t], b = recs[2 * t + 1];
    uint32_t P[4], N[4];
    if (old_layout) {
        P[0] = a.x; P[1] = a.y; P[2] = a.z; P[3] = a.w & 0xFu;
        N[0] = __funnelshift_r(a.w, b.x, 4);
        N[1] = __funnelshift_r(b.x, b.y, 4);
        N[2] = __funnelshift_r(b.y, b.z, 4);
        N[3] = (b.z >> 4) & 0xFu;
    } else {
        P[0] = a.x; P[1] = a.y; P[2] = a.z; P[3] = a.w & 0xFu;
        N[0] = b.x; N[1] = b.y; N[2] = b.z; N[3] = b.w & 0xFu;
    }
    const uint32_t pm[4] = {pair_mask.x, pair_mask.y, pair_mask.z, pair_mask.w};   // pairs the length has
#pragma unroll
    for (int k = 0; k < 4; ++k) { P[k] &= pm[k]; N[k] &= pm[k]; }
    const uint64_t e = entry_of(off, n_entries, t);
    store_record(recs, t, P, N, (uint32_t)e, (uint32_t)t - off[e], off[e + 1] - 1u - (uint32_t)t);
}

// synthetic ragged corpus: sub-fingerprint s of entry e is lbo_synth_entry's (oracle/lbad_oracle.c), entry e has
// lo + mix32(seed ^ 0x52414747 ^ e) % (hi - lo + 1) sub-fingerprints (the caller passes the prefix sums)
__device__ __forceinline__ uint32_t sl_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(kSlThreads) void synth_ragged_kernel(uint32_t seed, uint64_t first_entry, uint64_t n_entries,
                                                                  const uint32_t* __restrict__ off, uint64_t n_pos,
                                                                  uint32_t subfp_len, uint32_t* __restrict__ out) {
    const uint64_t p = (uint64_t)blockIdx.x * kSlThreads + threadIdx.x;
    if (p >= n_pos) return;
    uint64_t lo = 0, hi = n_entries;
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (off[mid] <= p) lo = mid; else hi = mid;
    }
    const uint64_t entry = first_entry + lo;
    const uint32_t s = (uint32_t)p - off[lo];
    const uint32_t key = sl_mix32(seed ^ sl_mix32((uint32_t)entry) ^ (uint32_t)(entry >> 32) * 0x632BE5ABu);
    const uint32_t pairs = (subfp_len + 1) / 2;
    uint32_t w[kPackedWords] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 4
    for (uint32_t pr = 0; pr < pairs; ++pr) {
        const uint32_t r = sl_mix32(key + (s * 1024u + pr) * 0x9E3779B1u);
        uint32_t pos = 0, neg = 0;
        if (r % 100u != 0u) {
            if ((r >> 8) & 1u) pos = 1; else neg = 1;
        }
        const uint32_t b = 2 * pr;
        if (b + 1 >= subfp_len) neg = 0;
        const uint32_t two = pos | (neg << 1);
#pragma unroll
        for (uint32_t k = 0; k < kPackedWords; ++k)
            if (k == (b >> 5)) w[k] |= two << (b & 31);
    }
    uint4* dst = reinterpret_cast<uint4*>(out + p * kPackedWords);
    dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
    dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

struct TriTable {
    std::mutex lock;
    float* d[kMaxDevices] = {};
};
TriTable g_tri;

// pair bits 0 .. ceil(limit / 2) - 1
uint4 pair_mask(uint32_t limit) {
    const uint32_t pairs = (limit + 1u) / 2u;
    uint32_t m[4];
    for (uint32_t w = 0; w < 4; ++w) {
        const uint32_t base = 32u * w;
        m[w] = pairs <= base ? 0u : (pairs - base >= 32u ? 0xFFFFFFFFu : ((1u << (pairs - base)) - 1u));
    }
    return make_uint4(m[0], m[1], m[2], m[3]);
}

uint4 sliding_range_mask(uint32_t subfp_len, uint32_t range) {
    return pair_mask(range < subfp_len ? range : subfp_len);      // Fp.m:155
}

}  // namespace

bool sliding_supported(uint32_t subfp_len) { return subfp_len >= 1 && subfp_len <= 2 * kTriPairs; }

uint32_t sliding_query_words(uint32_t n_query) { return kQHeader + n_query * kQWords; }

// the table of correctly rounded quotients hits / possible, row `possible` at possible (possible + 1) / 2
const float* sliding_tri_table() {
    const int dev = current_device();
    if (dev < 0 || dev >= kMaxDevices) return nullptr;
    std::lock_guard<std::mutex> g(g_tri.lock);
    if (g_tri.d[dev]) return g_tri.d[dev];
    std::vector<float> t(kTriSize, 0.0f);
    for (uint32_t p = 1; p <= kTriPairs; ++p)
        for (uint32_t h = 0; h <= p; ++h) t[p * (p + 1) / 2 + h] = (float)h / (float)p;   // Fp.m:175
    float* d = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), kTriSize * sizeof(float)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, t.data(), kTriSize * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d);
        return nullptr;
    }
    g_tri.d[dev] = d;
    return d;
}

// Host: the query block of the scan from unpacked Booleans (n_query x subfp_len): kQHeader reserved words, then 16
// words per sub-fingerprint
void build_sliding_query(const Boolean* bools, uint32_t n_query, uint32_t subfp_len, uint32_t range,
                         std::vector<uint32_t>& out) {
    out.assign((size_t)kQHeader + ((size_t)n_query + 1u) * kQWords, 0u);      // one zero sub-fingerprint of slack (fetch-ahead)
    const uint4 rm4 = sliding_range_mask(subfp_len, range);
    const uint32_t rm[4] = {rm4.x, rm4.y, rm4.z, rm4.w};
    const uint32_t pairs = (subfp_len + 1u) / 2u;
    for (uint32_t s = 0; s < n_query; ++s) {
        const Boolean* b = bools + (size_t)s * subfp_len;
        uint32_t* o = out.data() + kQHeader + (size_t)s * kQWords;
        for (uint32_t p = 0; p < pairs; ++p) {
            if (b[2 * p]) o[p >> 5] |= 1u << (p & 31);
            if (2 * p + 1 < subfp_len && b[2 * p + 1]) o[4 + (p >> 5)] |= 1u << (p & 31);
        }
        uint32_t possible = 0;
        for (uint32_t w = 0; w < 4; ++w) {
            o[8 + w] = (o[w] | o[4 + w]) & rm[w];
            possible += (uint32_t)__builtin_popcount(o[8 + w]);
        }
        o[12] = possible * (possible + 1u) / 2u;
        o[13] = possible;
    }
}

// d_off_new: ABSOLUTE record positions of the n_new new entries (n_new + 1 values, the first one = the position of slot 0)
hipError_t launch_pack_records(const uint32_t* d_slots, uint64_t n_new_pos, const uint32_t* d_off_new, uint64_t n_new,
                               uint32_t first_entry, uint4* d_recs, hipStream_t stream) {
    if (n_new_pos == 0) return hipSuccess;
    const uint64_t blocks = (n_new_pos + kSlThreads - 1) / kSlThreads;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pack_records_kernel, dim3((uint32_t)blocks), dim3(kSlThreads), 0, stream, d_slots, n_new_pos, d_off_new,
                       n_new, first_entry, d_recs);
    return hipGetLastError();
}

hipError_t launch_restamp_records(uint4* d_recs, const uint32_t* d_off, uint64_t n_entries, uint64_t n, uint32_t subfp_len,
                                  bool old_layout, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const uint64_t blocks = (n + kSlThreads - 1) / kSlThreads;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(restamp_records_kernel, dim3((uint32_t)blocks), dim3(kSlThreads), 0, stream, d_recs, d_off, n_entries, n,
                       old_layout ? 1u : 0u, pair_mask(subfp_len));
    return hipGetLastError();
}

hipError_t launch_synth_ragged(uint32_t seed, uint64_t first_entry, uint64_t n_entries, const uint32_t* d_off,
                               uint64_t n_pos, uint32_t subfp_len, uint32_t* d_out, hipStream_t stream) {
    if (n_pos == 0) return hipSuccess;
    const uint64_t blocks = (n_pos + kSlThreads - 1) / kSlThreads;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(synth_ragged_kernel, dim3((uint32_t)blocks), dim3(kSlThreads), 0, stream, seed, first_entry,
                       n_entries, d_off, n_pos, subfp_len, d_out);
    return hipGetLastError();
}

// ---- the plan and the scan ---------------------------------------------------------------------------------------------
// The systolic scan (compare_short_kernel) takes
//   * queries of up to LBAD_SHORT_QUERY = 7 sub-fingerprints: one record per lane, HBM-bound (0.26 ms at 5 against 0.31 of the
//     task kernel), and
//   * corpora whose LONGEST entry has at most 15 records (the records' place fields saturate at 15, which bounds the reach
//     of a window at 14), whatever the query: "B" work on short entries costs the task kernel a window fill per n steps
//     (4 M entries of 8..15 against a query of 100: 2.1 ms systolic, 15.5 ms task kernel).
// Queries of 8..15 against longer entries went to the systolic scan's four-records-per-lane form until round 5; with the
// whole-line window fill the task kernel is 12-15 % faster there (1 M entries of 20..70: 0.315 / 0.336 / 0.352 ms at
// 8 / 12 / 15 against 0.370 / 0.382 / 0.404) and its batches of eight 13-30 %.
#ifndef LBAD_SHORT_QUERY
#define LBAD_SHORT_QUERY 7
#endif
constexpr uint32_t kShortEntries = 15;
bool sliding_short(uint32_t n_query, uint32_t ne_max) { return n_query <= LBAD_SHORT_QUERY || ne_max <= kShortEntries; }
// A BATCH of such queries goes through compare_short_multi_kernel (defined above; needs an entry longer than the query)
bool sliding_multi(uint32_t n_query, uint32_t ne_max) { return n_query <= LBAD_SHORT_MULTI_MAX && ne_max > n_query; }

static void sliding_variant(uint32_t subfp_len, uint32_t n_query, uint32_t range, uint32_t n_q, bool& full, bool& qlds, uint32_t& dyn_lds) {
    const uint4 rm = sliding_range_mask(subfp_len, range);
    const uint4 all = pair_mask(subfp_len);
    full = rm.x == all.x && rm.y == all.y && rm.z == all.z && rm.w == all.w;
    qlds = n_q > 1 || n_query <= kQueryLds;
    dyn_lds = qlds ? n_q * (n_query + 1u) * kQWords * 4u : 0u;
}

#ifndef LBAD_MULTI_THREADS
#define LBAD_MULTI_THREADS 1024
#endif
#ifndef LBAD_MULTI_THREADS8
#define LBAD_MULTI_THREADS8 768
#endif
#ifndef LBAD_MULTI_MAX
#define LBAD_MULTI_MAX 4u
#endif
constexpr int kMultiThreads = LBAD_MULTI_THREADS;  // two or four queries per pass: sixteen waves like the single scan (126 registers)
constexpr int kMultiThreads8 = LBAD_MULTI_THREADS8; // eight queries per pass: twelve waves, 168 registers per lane
constexpr uint32_t kMultiLdsWords = 7000;           // dynamic LDS a launch of several queries may ask for (28 KB next to 131 KB of tables and queues)

// How many of `n_left` queries of n_query sub-fingerprints ONE launch takes: the systolic scan of short queries up to
// eight, the task scan four or two while their blocks fit the LDS next to the tables.
uint32_t sliding_queries_per_launch(uint32_t n_query, uint32_t ne_max, uint32_t n_left) {
    if (n_left <= 1) return n_left;
    if (sliding_multi(n_query, ne_max)) return n_left >= 8 ? 8u : (n_left >= 4 ? 4u : 2u);    // compare_short_multi_kernel
    if (sliding_short(n_query, ne_max)) {
        // one record per lane (windows of up to seven records): eight queries side by side; four records per lane: four (eight
        // would need 213 registers -- two waves per SIMD -- and gain nothing over two launches of four)
        const uint32_t look = (n_query < ne_max ? n_query : ne_max) - 1u;
        const uint32_t most = look <= 6u ? 8u : 4u;
        return n_left >= most ? most : (n_left >= 4 ? 4u : 2u);
    }
    uint32_t g = n_left >= LBAD_MULTI_MAX ? LBAD_MULTI_MAX : (n_left >= 4 ? 4u : 2u);
    while (g > 1 && (uint64_t)g * (n_query + 1u) * kQWords > kMultiLdsWords) g >>= 1;
    return g;
}

// Shape of a scan: one workgroup per CU, each with 1 / grid of the tasks of either kind (whole entries).
SlideShape sliding_shape(uint64_t tasks_a, uint64_t tasks_b, uint32_t n_q) {
    SlideShape sh;
    const uint64_t waves = n_q == 8 ? kMultiThreads8 / 64 : (n_q > 1 ? kMultiThreads / 64 : kScanWaves);
    const uint64_t passes = (tasks_a + 63) / 64 + (tasks_b + 63) / 64;
    const uint64_t want = (passes + waves - 1) / waves;
    uint64_t cap = (uint64_t)device_cu_count() * kScanPerCu;
    if (cap > kSlideMaxGrid) cap = kSlideMaxGrid;
    sh.grid = (uint32_t)(want < cap ? (want ? want : 1) : cap);
    auto chunk = [&](uint64_t tasks) -> uint32_t {
        if (tasks == 0) return 0u;
        const uint64_t per = (tasks + sh.grid - 1) / sh.grid;
        return (uint32_t)(per ? per : 1);
    };
    sh.chunk_a = chunk(tasks_a);
    sh.chunk_b = chunk(tasks_b);
    return sh;
}

// words of a plan buffer for a corpus of `capacity` entries: starts_a, starts_b (kSlideMaxGrid + 1 each), block sums
// (2 per 1024 entries)
size_t sliding_plan_words(uint64_t capacity) {
    return 2 * (size_t)(kSlideMaxGrid + 1) + 2 * (size_t)((capacity + kPlanPerBlock - 1) / kPlanPerBlock) + 8;
}

// Where every workgroup's run of entries starts, for queries of n_query sub-fingerprints: three small launches (block
// sums of the entries' task counts, their scan, the boundaries), about 15 us for 1 M entries; the corpus keeps the plan
// until the query length or the entries change.
hipError_t launch_sliding_plan(const uint32_t* d_off, uint64_t n_entries, uint32_t n_query, uint32_t b_min, const SlideShape& sh,
                               uint32_t* d_plan, hipStream_t stream) {
    if (n_entries == 0) return hipSuccess;
    uint32_t* starts_a = d_plan;
    uint32_t* starts_b = starts_a + (kSlideMaxGrid + 1);
    uint32_t* sums = starts_b + (kSlideMaxGrid + 1);
    const uint64_t nb = (n_entries + kPlanPerBlock - 1) / kPlanPerBlock;
    if (nb > 0x7fffffffull || n_entries > 0xFFFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(slide_plan_sums_kernel, dim3((uint32_t)nb), dim3(kSlThreads), 0, stream, d_off, n_entries, n_query, b_min, sums);
    hipLaunchKernelGGL(slide_plan_scan_kernel, dim3(1), dim3(1024), 0, stream, sums, (uint32_t)nb, (uint32_t)n_entries, sh.grid,
                       sh.chunk_a, sh.chunk_b, starts_a, starts_b);
    hipLaunchKernelGGL(slide_plan_final_kernel, dim3((uint32_t)nb), dim3(kSlThreads), 0, stream, d_off, n_entries, n_query, b_min, sums,
                       sh.chunk_a, sh.chunk_b, starts_a, starts_b);
    return hipGetLastError();
}

// tasks_a / tasks_b: groups of four sliding offsets over the entries longer than / not longer than the query (from the
// host's histogram of entry lengths).  zero_rec: index of an all-zero record behind the stored ones.  d_score_bits
// (optional, one query only; n_entries words) must be zero on entry and receives the float bits of every entry's match.
// scan: the queries (build_sliding_query blocks without their header, one after the other) and where the keys go.
hipError_t launch_compare_sliding(const uint4* d_recs, uint64_t n_pos, const uint32_t* d_off, uint64_t n_entries, uint32_t ne_max,
                                  uint32_t zero_rec, uint64_t tasks_a, uint64_t tasks_b, const SlideShape& sh, const uint32_t* d_plan,
                                  uint32_t subfp_len, const SlideScan& scan, uint32_t n_query, uint32_t range,
                                  uint64_t index_base, unsigned int* d_score_bits, hipStream_t stream, bool bound_pruning,
                                  float prune_from, uint32_t b_min) {
    const uint32_t n_q = scan.n_q;
    if (n_entries == 0 || n_query == 0 || tasks_a + tasks_b == 0 || n_q == 0) return hipErrorInvalidValue;   // (the caller clears the keys itself)
    if (tasks_a > 0xFFFFFFFFull || tasks_b > 0xFFFFFFFFull) return hipErrorInvalidValue;    // the plan counts in 32 bits
    if (n_query >= (1u << 20)) return hipErrorInvalidValue;                                  // (run_pass: 32-bit lane offsets)
    if (n_q > 1 && d_score_bits) return hipErrorInvalidValue;
    const float* tri = sliding_tri_table();
    if (!tri) return hipErrorOutOfMemory;
    ScanOut out;
    out.acc = scan.d_acc; out.ticket = scan.d_ticket; out.keys = scan.d_keys;
    for (int i = 0; i < 8; ++i) out.pos[i] = scan.key_pos[i];
    // the systolic scan: 64 K records per wave and chunk (K = 4 records per lane once a window reaches back more than six
    // records: the overlap of consecutive chunks stays a small part of a chunk); only_upto: see compare_short_kernel
    auto run_short = [&](uint32_t look, uint32_t only_upto) -> hipError_t {
        if (!scan.d_queries) return hipErrorInvalidValue;
        const uint32_t K = look <= 6u ? 1u : 4u;
        const uint32_t step = 64u * K - look;
        const uint64_t span = 64ull * K;
        const uint64_t n_chunks = n_pos <= span ? 1u : (n_pos - span + step - 1u) / step + 1u;
        const uint64_t want = (n_chunks + (kSlThreads / 64) - 1) / (kSlThreads / 64);
        const uint64_t cap = (uint64_t)device_cu_count() * 6u;                   // 21 KB of LDS per workgroup
        const uint32_t grid = (uint32_t)(want < cap ? want : cap);
        const uint4 rm4 = sliding_range_mask(subfp_len, range);
#define LBAD_SHORT(KK, QQ)                                                                                                    \
    hipLaunchKernelGGL((compare_short_kernel<KK, QQ>), dim3(grid), dim3(kSlThreads), 0, stream, d_recs, n_pos, scan.d_queries, \
                       n_query, step, n_chunks, rm4, tri, index_base, d_score_bits, out, only_upto)
        if (K == 1) {
            if (n_q == 1) LBAD_SHORT(1, 1); else if (n_q == 2) LBAD_SHORT(1, 2); else if (n_q == 4) LBAD_SHORT(1, 4);
            else if (n_q == 8) LBAD_SHORT(1, 8); else return hipErrorInvalidValue;
        } else {
            if (n_q == 1) LBAD_SHORT(4, 1); else if (n_q == 2) LBAD_SHORT(4, 2); else if (n_q == 4) LBAD_SHORT(4, 4);
            else if (n_q == 8) LBAD_SHORT(4, 8); else return hipErrorInvalidValue;
        }
#undef LBAD_SHORT
        return hipGetLastError();
    };
    // several queries of up to seven sub-fingerprints: the entries longer than the query through compare_short_multi_kernel
    // (four records per lane, query words in vector registers), the others -- where the corpus has any (tasks_b counts their
    // offsets) -- through the systolic scan above in its only_upto mode, which maxes into the same keys
    auto run_short_multi = [&]() -> hipError_t {
        if (!scan.d_queries) return hipErrorInvalidValue;
        const uint32_t look = n_query - 1u;
        const uint32_t step = 64u * kShortMultiK - look;
        const uint64_t span = 64ull * kShortMultiK;
        const uint64_t n_chunks = n_pos <= span ? 1u : (n_pos - span + step - 1u) / step + 1u;
        // one workgroup of sixteen waves per CU, each with a contiguous run of chunks its waves claim from an LDS cursor
        const uint64_t waves = kSmThreads / 64;
        const uint64_t want = (n_chunks + waves - 1) / waves;
        const uint64_t cap = (uint64_t)device_cu_count();
        const uint32_t grid = (uint32_t)(want < cap ? want : cap);
        const uint64_t per_group = (n_chunks + grid - 1) / grid;
        if (per_group >= 0xFFFFFFFFull) return hipErrorInvalidValue;
        const uint4 rm4 = sliding_range_mask(subfp_len, range);
#define LBAD_SHORT_MULTI(QQ, NN)                                                                                                  \
    hipLaunchKernelGGL((compare_short_multi_kernel<QQ, NN>), dim3(grid), dim3(kSmThreads), 0, stream, d_recs, n_pos, scan.d_queries, \
                       step, n_chunks, per_group, rm4, index_base, out)
#if LBAD_SHORT_MULTI_MAX > 7
#define LBAD_SHORT_MULTI_MORE(QQ)                                                                                                 \
        case 8: LBAD_SHORT_MULTI(QQ, 8); break;                                                                                   \
        case 9: LBAD_SHORT_MULTI(QQ, 9); break;                                                                                   \
        case 10: LBAD_SHORT_MULTI(QQ, 10); break;                                                                                 \
        case 11: LBAD_SHORT_MULTI(QQ, 11); break;                                                                                 \
        case 12: LBAD_SHORT_MULTI(QQ, 12); break;
#else
#define LBAD_SHORT_MULTI_MORE(QQ)
#endif
#define LBAD_SHORT_MULTI_N(QQ)                                                                                                    \
    switch (n_query) {                                                                                                            \
        case 1: LBAD_SHORT_MULTI(QQ, 1); break;                                                                                   \
        case 2: LBAD_SHORT_MULTI(QQ, 2); break;                                                                                   \
        case 3: LBAD_SHORT_MULTI(QQ, 3); break;                                                                                   \
        case 4: LBAD_SHORT_MULTI(QQ, 4); break;                                                                                   \
        case 5: LBAD_SHORT_MULTI(QQ, 5); break;                                                                                   \
        case 6: LBAD_SHORT_MULTI(QQ, 6); break;                                                                                   \
        case 7: LBAD_SHORT_MULTI(QQ, 7); break;                                                                                   \
        LBAD_SHORT_MULTI_MORE(QQ)                                                                                                 \
        default: return hipErrorInvalidValue;                                                                                     \
    }
        if (n_q == 2) { LBAD_SHORT_MULTI_N(2) } else if (n_q == 4) { LBAD_SHORT_MULTI_N(4) } else if (n_q == 8) { LBAD_SHORT_MULTI_N(8) }
        else return hipErrorInvalidValue;
#undef LBAD_SHORT_MULTI_N
#undef LBAD_SHORT_MULTI
        const hipError_t launched = hipGetLastError();
        if (launched != hipSuccess || tasks_b == 0) return launched;
        return run_short(look, n_query);
    };
    if (n_q > 1 && sliding_multi(n_query, ne_max) && tasks_a != 0 && !d_score_bits) return run_short_multi();
    if (sliding_short(n_query, ne_max)) return run_short((n_query < ne_max ? n_query : ne_max) - 1u, 0u);
    SlideArgs a;
    a.index_base = index_base; a.n_entries = n_entries; a.nq = n_query; a.zero_rec = zero_rec;
    const uint4 rm = sliding_range_mask(subfp_len, range);
    a.rm[0] = rm.x; a.rm[1] = rm.y; a.rm[2] = rm.z; a.rm[3] = rm.w;
    auto dense = [&](uint64_t tasks) -> uint32_t {
        if (tasks == 0) return 8u;
        const uint64_t e = (64ull * LBAD_CLAIM_PASSES * n_entries + tasks - 1) / tasks;
        return (uint32_t)(e < 8ull ? 8ull : (e > 4096ull ? 4096ull : e));
    };
    a.dense_a = dense(tasks_a);
    a.dense_b = dense(tasks_b);
    a.prune = (bound_pruning && d_score_bits == nullptr && n_q == 1 && n_query <= kPruneMaxQuery) ? 1u : 0u;
    a.prune_from = prune_from;
    a.b_min = b_min;
    bool full, qlds;
    uint32_t dyn_lds;
    sliding_variant(subfp_len, n_query, range, n_q, full, qlds, dyn_lds);
    static QueryArg qa_zero = {};
    const QueryArg* qa = &qa_zero;
    QueryArg qa_local;
    a.q_in_args = 0;
    if (n_q == 1 && qlds && scan.h_query && n_query <= kSlideQueryArgSubs) {
        std::memcpy(qa_local.w, scan.h_query, (size_t)(n_query + 1u) * kQWords * 4u);
        qa = &qa_local;
        a.q_in_args = 1;
    } else if (!scan.d_queries) {
        return hipErrorInvalidValue;
    }
    const uint32_t* starts_a = d_plan;
    const uint32_t* starts_b = starts_a + (kSlideMaxGrid + 1);
    const uint32_t* q = scan.d_queries;
#define LBAD_SLIDE(FULL, QL, QQ, TT)                                                                                          \
    hipLaunchKernelGGL((compare_sliding_kernel<FULL, false, QL, QQ, TT>), dim3(sh.grid), dim3(TT), dyn_lds, stream, d_recs, d_off, \
                       q, starts_a, starts_b, tri, d_score_bits, out, a, *qa)
    if (n_q == 1) {
        if (full) { if (qlds) LBAD_SLIDE(true, true, 1, kScanThreads); else LBAD_SLIDE(true, false, 1, kScanThreads); }
        else { if (qlds) LBAD_SLIDE(false, true, 1, kScanThreads); else LBAD_SLIDE(false, false, 1, kScanThreads); }
    } else if (n_q == 2) {
        if (full) LBAD_SLIDE(true, true, 2, kMultiThreads); else LBAD_SLIDE(false, true, 2, kMultiThreads);
    } else if (n_q == 4) {
        if (full) LBAD_SLIDE(true, true, 4, kMultiThreads); else LBAD_SLIDE(false, true, 4, kMultiThreads);
#if LBAD_MULTI_MAX >= 8
    } else if (n_q == 8) {
        if (full) LBAD_SLIDE(true, true, 8, kMultiThreads8); else LBAD_SLIDE(false, true, 8, kMultiThreads8);
#endif
    } else {
        return hipErrorInvalidValue;
    }
#undef LBAD_SLIDE
    const hipError_t launched = hipGetLastError();
    if (launched != hipSuccess || b_min == 0) return launched;
    // The corpus is split: "B" entries of fewer than b_min sub-fingerprints were no tasks above (an entry of n records
    // costs the task kernel a pass of n_query steps per four offsets whatever n is: 15.5 ms against 2.1 for 4 M entries of
    // 8..15 and a query of 100).  The systolic scan takes them -- every record once, n_query steps per record, nothing for
    // chunks without such an entry -- and maxes into the keys the task kernel has just written (same stream).
    return run_short(b_min - 2u, b_min - 1u);
}

#if defined(LBAD_SLIDE_PROF) || defined(LBAD_SLIDE_STAMPS)
extern "C" int LBAudioDetectiveDebugSlideTimes(unsigned long long* out, int n_words) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_slide_times), sizeof(unsigned long long) * (size_t)n_words) == hipSuccess ? 0 : 1;
}
extern "C" int LBAudioDetectiveDebugSlideTimesReset() {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_slide_times)) != hipSuccess) return 1;
    return hipMemset(p, 0, sizeof(unsigned long long) * 1024 * 16 * 8) == hipSuccess ? 0 : 1;
}
#endif
#ifdef LBAD_SLIDE_PROF
extern "C" int LBAudioDetectiveDebugSlidePasses(unsigned int* out, int n_words) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_slide_passes), sizeof(unsigned int) * (size_t)n_words) == hipSuccess ? 0 : 1;
}
extern "C" int LBAudioDetectiveDebugSlideProfile(unsigned long long* out16, int reset) {
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_slide_prof), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[16] = {};
        z[11] = ~0ull;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_slide_prof), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

}  // namespace lbad
