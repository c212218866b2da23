// k_sliding.hip -- corpus scan for entries of ANY length: the sliding compare of
// LBAudioDetectiveFingerprintCompareToFingerprint (LBAudioDetectiveFingerprint.m:119-149) with every entry of a
// ragged corpus as the second argument and the query as the first, i.e. the best-match loop of
// LBAudioDetectiveTests/LBAudioDetectiveTests.m:57-91 (one original against sequences of other lengths) as ONE
// launch.
//
// The corpus is a stream of sub-fingerprint RECORDS, 32 bytes each, entries back to back:
//   bits   0.. 99  P   first Boolean of pair p at bit p        (pairs = ceil(length / 2) <= 100)
//   bits 100..199  N   second Boolean of pair p at bit 100 + p
//   bits 200..211  i   index of the sub-fingerprint inside its entry, saturated at 4095
//   bits 212..223  r   sub-fingerprints that follow it inside its entry, saturated at 4095
//   bits 224..255  e   index of the entry
// so a lane that loads one record knows everything about its place, no side table is read by the scan and the
// loads are two aligned, fully coalesced dwordx4 per lane.  25 of the 32 bytes are the reference's information
// (25 B per sub-fingerprint).
//
// Per pair of sub-fingerprints (A = the longer fingerprint's, B = the other's; Fp.m:151-176):
//   NZ = (PA | NA) & RANGE,  possible = popc(NZ),  hits = popc(NZ & ~((PA ^ PB) | (NA ^ NB)))
//   ratio = hits / possible (0 when possible == 0) -- taken from a triangular table of the 5151 correctly rounded
//   quotients in LDS instead of an IEEE division per pair.
//
// Systolic evaluation.  Lane l of a wave holds record base + l in registers; the query's sub-fingerprints a = 0, 1,
// ... are wave-uniform (scalar loads).  In step a every lane computes ratio(a, l) and adds it to an accumulator
// that moves one lane to the right per step (`v_add_f32 ... wave_shr:1`), so an accumulator follows one diagonal
// l - a = const: exactly the terms of one sliding offset, added in the reference's order (Fp.m:139-142).
//   entry longer than the query ("A" lanes): a diagonal starts in step 0 in every lane and is complete after the
//     last step; it is an offset of the entry iff it stayed inside the entry (i >= n_query - 1).
//   entry not longer than the query ("B" lanes): a diagonal starts whenever it enters the entry's first lane
//     (i == 0) and is complete when it leaves the last one (r == 0); that lane keeps the maximum over the steps.
// Diagonals that did not start properly carry -inf.  max over offsets commutes with the division by n2 (a correctly
// rounded division by a positive constant is monotonic), so there is ONE division per lane and chunk
// (Fp.m:144).  Consecutive chunks of 64 records overlap by min(n_query, longest entry) - 1 records; a window
// inside the overlap is evaluated twice with the same result.
#include "internal.hpp"

#include <cmath>
#include <cstring>
#include <mutex>

namespace lbad {
namespace {

constexpr int kSlThreads = 256;
constexpr uint32_t kTriPairs = 100;
constexpr uint32_t kTriSize = (kTriPairs + 1) * (kTriPairs + 2) / 2;   // 5151 quotients
constexpr uint32_t kQWords = 16;   // per query sub-fingerprint: P[4] N[4] NZ[4] tri-base possible - -

struct Rec {
    uint32_t P[4], N[4];
    uint32_t isat, rem, idx;
};

__device__ __forceinline__ Rec unpack_rec(const uint4 a, const uint4 b) {
    Rec r;
    r.P[0] = a.x; r.P[1] = a.y; r.P[2] = a.z; r.P[3] = a.w & 0xFu;
    r.N[0] = __funnelshift_r(a.w, b.x, 4);
    r.N[1] = __funnelshift_r(b.x, b.y, 4);
    r.N[2] = __funnelshift_r(b.y, b.z, 4);
    r.N[3] = (b.z >> 4) & 0xFu;
    r.isat = (b.z >> 8) & 0xFFFu;
    r.rem = b.z >> 20;
    r.idx = b.w;
    return r;
}

__device__ __forceinline__ unsigned long long sl_key(float score, uint64_t global_index) {
    return ((unsigned long long)__float_as_uint(score) << 32) |
           (unsigned long long)(0xFFFFFFFFu - (uint32_t)global_index);
}

// value of lane l - 1 (lane 0 receives 0; whatever enters a chunk from the left belongs to a window that is not
// wholly inside the chunk and is never used)
__device__ __forceinline__ float from_left_lane(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138 /* wave_shr:1 */, 0xF, 0xF, true));
}

constexpr int kNegInf = (int)0xFF800000u;   // -inf; as a signed integer it sorts below the bits of every sum >= 0

// One chunk = 64 K consecutive records; lane l holds records K l .. K l + K - 1, so a diagonal moves from register
// set k - 1 to set k inside the lane and crosses to the next lane only from set K - 1 to set 0.
// MODE 0: every entry in the chunk is longer than the query; 1: none is; 2: mixed.
template <int K, int MODE>
__device__ __forceinline__ void run_steps(const uint32_t (&P)[K][4], const uint32_t (&N)[K][4], const uint32_t (&nz)[K][4],
                                          const uint32_t (&tri)[K], const bool (&case_a)[K], const bool (&start_b)[K],
                                          const uint32_t* __restrict__ q, uint32_t nq, const float* s_tri,
                                          float (&acc)[K], int (&smax)[K]) {
    // MODE 2: the mask of a cell is the entry's NZ in A lanes and the query's in B lanes:
    //   m = sel_e & (nz_q | sel_a)   with sel_e = A ? nz_e : ~0,  sel_a = A ? ~0 : 0     (one v_bitop3)
    uint32_t sel_e[K][4], sel_a[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        acc[k] = MODE == 0 ? 0.0f : __int_as_float(kNegInf);   // MODE 0: step 0 adds to the zeros, no reset needed
        smax[k] = kNegInf;
        sel_a[k] = case_a[k] ? 0xFFFFFFFFu : 0u;
#pragma unroll
        for (int w = 0; w < 4; ++w) sel_e[k][w] = case_a[k] ? nz[k][w] : 0xFFFFFFFFu;
    }
    for (uint32_t a = 0; a < nq; ++a) {
        const uint32_t* __restrict__ qa = q + (size_t)a * kQWords;
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
        const float in0 = from_left_lane(acc[K - 1]);
#pragma unroll
        for (int k = K - 1; k >= 0; --k) {
            float sh = k ? acc[k - 1] : in0;
            if (MODE == 1) sh = start_b[k] ? 0.0f : sh;
            if (MODE == 2) sh = (start_b[k] || (a == 0 && case_a[k])) ? 0.0f : sh;
            acc[k] = __fadd_rn(sh, ratio[k]);
            if (MODE != 0) smax[k] = max(smax[k], __float_as_int(acc[k]));
        }
    }
}

template <int K>
__global__ __launch_bounds__(kSlThreads) void compare_sliding_kernel(
    const uint4* __restrict__ recs, uint64_t n_pos, const uint32_t* __restrict__ q, uint32_t nq, uint32_t chunk_step,
    uint64_t n_chunks, uint4 range_mask, const float* __restrict__ tri_tbl, uint64_t index_base,
    unsigned int* __restrict__ score_bits, unsigned long long* __restrict__ key_out) {
    __shared__ float s_tri[kTriSize];
    __shared__ unsigned long long s_k[kSlThreads / 64];
    for (uint32_t i = threadIdx.x; i < kTriSize; i += kSlThreads) s_tri[i] = tri_tbl[i];
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = (uint64_t)blockIdx.x * (kSlThreads / 64) + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (kSlThreads / 64);
    const uint32_t rm[4] = {range_mask.x, range_mask.y, range_mask.z, range_mask.w};
    unsigned long long best = 0ull;

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
            const uint32_t ne = r.isat + r.rem + 1u;       // saturated; exact whenever it is <= nq
            case_a[k] = ne > nq;                           // the entry is the longer side (Fp.m:123-131)
            n2[k] = case_a[k] ? nq : ne;
            start_b[k] = !case_a[k] && r.isat == 0u;
            some_a |= case_a[k] && inb[k];
            some_b |= !case_a[k] && inb[k];
        }
        const bool any_a = __ballot(some_a) != 0ull, any_b = __ballot(some_b) != 0ull;

        float acc[K];
        int smax[K];
        if (!any_b) run_steps<K, 0>(P, N, nz, tri, case_a, start_b, q, nq, s_tri, acc, smax);
        else if (!any_a) run_steps<K, 1>(P, N, nz, tri, case_a, start_b, q, nq, s_tri, acc, smax);
        else run_steps<K, 2>(P, N, nz, tri, case_a, start_b, q, nq, s_tri, acc, smax);

        // a record closes a window iff the window lies inside its entry AND inside this chunk.  The exact
        // division (Fp.m:144) runs only where the sum can reach the lane's best so far.
        const float thr = __uint_as_float((uint32_t)(best >> 32)) * 0.99999f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const bool closes = case_a[k] ? (isat[k] >= nq - 1u) : (rem[k] == 0u);
            const bool valid = inb[k] && closes && lane * K + k >= n2[k] - 1u;
            const float s = case_a[k] ? acc[k] : __int_as_float(smax[k]);
            const float n2f = (float)n2[k];
            const bool need = valid && (score_bits != nullptr || s >= thr * n2f);
            if (__ballot(need) != 0ull) {
                if (need) {
                    const float cand = __fdiv_rn(s, n2f);
                    const float match = (0.0f < cand) ? cand : 0.0f;         // MAX(match, cand) from match = 0
                    if (score_bits) atomicMax(&score_bits[idx[k]], __float_as_uint(match));
                    const unsigned long long key = sl_key(match, index_base + idx[k]);
                    best = key > best ? key : best;
                }
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(best, off, 64);
        best = o > best ? o : best;
    }
    if (lane == 0) s_k[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long m = s_k[0];
        for (int i = 1; i < kSlThreads / 64; ++i) m = s_k[i] > m ? s_k[i] : m;
        if (m) atomicMax(key_out, m);
    }
}

// Fallback for a long query against long entries (overlap of 64 records or more): one workgroup per entry,
// one thread per sliding offset, true divisions.
__global__ __launch_bounds__(kSlThreads) void compare_ragged_long_kernel(
    const uint4* __restrict__ recs, const uint32_t* __restrict__ off, uint64_t n_entries, const uint32_t* __restrict__ q,
    uint32_t nq, uint4 range_mask, uint64_t index_base, unsigned int* __restrict__ score_bits,
    unsigned long long* __restrict__ key_out) {
    __shared__ unsigned int s_best[kSlThreads / 64];
    const uint32_t rm[4] = {range_mask.x, range_mask.y, range_mask.z, range_mask.w};
    unsigned long long best_key = 0ull;
    for (uint64_t e = blockIdx.x; e < n_entries; e += gridDim.x) {
        const uint32_t p0 = off[e];
        const uint32_t ne = off[e + 1] - p0;
        const bool case_a = ne > nq;
        const uint32_t n1 = case_a ? ne : nq, n2 = case_a ? nq : ne;
        unsigned int best = 0u;
        for (uint32_t o = threadIdx.x; o + n2 <= n1; o += kSlThreads) {
            float sum = 0.0f;
            for (uint32_t i = 0; i < n2; ++i) {
                const uint32_t es = case_a ? i + o : i;     // entry sub-fingerprint
                const uint32_t qs = case_a ? i : i + o;     // query sub-fingerprint
                const Rec r = unpack_rec(recs[2 * (uint64_t)(p0 + es)], recs[2 * (uint64_t)(p0 + es) + 1]);
                const uint32_t* qa = q + (size_t)qs * kQWords;
                uint32_t hits = 0, possible = 0;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const uint32_t t = (r.P[w] ^ qa[w]) | (r.N[w] ^ qa[4 + w]);
                    const uint32_t nz = case_a ? (r.P[w] | r.N[w]) & rm[w] : qa[8 + w];
                    possible += __popc(nz);
                    hits += __popc(nz & ~t);
                }
                sum = __fadd_rn(sum, possible ? __fdiv_rn((float)hits, (float)possible) : 0.0f);
            }
            const unsigned int bits = __float_as_uint(__fdiv_rn(sum, (float)n2));
            best = bits > best ? bits : best;
        }
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) {
            const unsigned int v = __shfl_xor(best, sh, 64);
            best = v > best ? v : best;
        }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) s_best[threadIdx.x >> 6] = best;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned int m = s_best[0];
            for (int i = 1; i < kSlThreads / 64; ++i) m = s_best[i] > m ? s_best[i] : m;
            if (score_bits) score_bits[e] = m;
            const unsigned long long k = ((unsigned long long)m << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)(index_base + e));
            best_key = k > best_key ? k : best_key;
        }
    }
    if (threadIdx.x == 0 && best_key) atomicMax(key_out, best_key);
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

// packed sub-fingerprints (8-word slots, Boolean b at bit b) of whole entries -> records.  off: ABSOLUTE record
// positions of the new entries (n_new + 1 values); slot t becomes record off[0] + t.
__global__ __launch_bounds__(kSlThreads) void pack_records_kernel(const uint32_t* __restrict__ slots, uint64_t n_new_pos,
                                                                  const uint32_t* __restrict__ off, uint64_t n_new,
                                                                  uint32_t first_entry, uint4* __restrict__ recs) {
    const uint64_t t = (uint64_t)blockIdx.x * kSlThreads + threadIdx.x;
    if (t >= n_new_pos) return;
    const uint64_t p = (uint64_t)off[0] + t;
    // entry with off[e] <= p < off[e + 1]
    uint64_t lo = 0, hi = n_new;
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (off[mid] <= p) lo = mid; else hi = mid;
    }
    const uint32_t i = (uint32_t)p - off[lo];
    const uint32_t rem = off[lo + 1] - 1u - (uint32_t)p;
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
    P[3] &= 0xFu;
    N[3] &= 0xFu;
    const uint32_t isat = i < 4095u ? i : 4095u, rsat = rem < 4095u ? rem : 4095u;
    uint4 ra, rb;
    ra.x = P[0]; ra.y = P[1]; ra.z = P[2];
    ra.w = P[3] | (N[0] << 4);
    rb.x = (N[0] >> 28) | (N[1] << 4);
    rb.y = (N[1] >> 28) | (N[2] << 4);
    rb.z = (N[2] >> 28) | (N[3] << 4) | (isat << 8) | (rsat << 20);
    rb.w = first_entry + (uint32_t)lo;
    recs[2 * p] = ra;
    recs[2 * p + 1] = rb;
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

uint4 sliding_range_mask(uint32_t subfp_len, uint32_t range) {
    const uint32_t lim = range < subfp_len ? range : subfp_len;      // Fp.m:155
    const uint32_t pairs = (lim + 1u) / 2u;
    uint32_t m[4];
    for (uint32_t w = 0; w < 4; ++w) {
        const uint32_t base = 32u * w;
        m[w] = pairs <= base ? 0u : (pairs - base >= 32u ? 0xFFFFFFFFu : ((1u << (pairs - base)) - 1u));
    }
    return make_uint4(m[0], m[1], m[2], m[3]);
}

}  // namespace

bool sliding_supported(uint32_t subfp_len) { return subfp_len >= 1 && subfp_len <= 2 * kTriPairs; }

uint32_t sliding_query_words(uint32_t n_query) { return n_query * kQWords; }

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

// Host: the query block of the scan from unpacked Booleans (n_query x subfp_len)
void build_sliding_query(const Boolean* bools, uint32_t n_query, uint32_t subfp_len, uint32_t range,
                         std::vector<uint32_t>& out) {
    out.assign((size_t)n_query * kQWords, 0u);
    const uint4 rm4 = sliding_range_mask(subfp_len, range);
    const uint32_t rm[4] = {rm4.x, rm4.y, rm4.z, rm4.w};
    const uint32_t pairs = (subfp_len + 1u) / 2u;
    for (uint32_t s = 0; s < n_query; ++s) {
        const Boolean* b = bools + (size_t)s * subfp_len;
        uint32_t* o = out.data() + (size_t)s * kQWords;
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

hipError_t launch_pack_records(const uint32_t* d_slots, uint64_t n_new_pos, const uint32_t* d_off_new, uint64_t n_new,
                               uint32_t first_entry, uint4* d_recs_at_first, hipStream_t stream) {
    if (n_new_pos == 0) return hipSuccess;
    const uint64_t blocks = (n_new_pos + kSlThreads - 1) / kSlThreads;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pack_records_kernel, dim3((uint32_t)blocks), dim3(kSlThreads), 0, stream, d_slots, n_new_pos,
                       d_off_new, n_new, first_entry, d_recs_at_first);
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

// d_off: n_entries + 1 record positions; ne_max: the longest entry.  d_score_bits (optional, n_entries words)
// must be zero on entry and receives the float bits of every entry's match; *d_key is max-ed.
hipError_t launch_compare_sliding(const uint4* d_recs, uint64_t n_pos, const uint32_t* d_off, uint64_t n_entries,
                                  uint32_t ne_max, uint32_t subfp_len, const uint32_t* d_qblk, uint32_t n_query,
                                  uint32_t range, uint64_t index_base, unsigned int* d_score_bits,
                                  unsigned long long* d_key, hipStream_t stream) {
    if (n_entries == 0 || n_pos == 0 || n_query == 0) return hipSuccess;
    const uint4 rm = sliding_range_mask(subfp_len, range);
    const uint32_t look = (n_query < ne_max ? n_query : ne_max) - 1u;   // records a window reaches back
    const int cus = device_cu_count();
    if (look >= 64u) {
        const uint64_t cap = (uint64_t)cus * 8u;
        const uint32_t grid = (uint32_t)(n_entries < cap ? n_entries : cap);
        hipLaunchKernelGGL(compare_ragged_long_kernel, dim3(grid), dim3(kSlThreads), 0, stream, d_recs, d_off, n_entries,
                           d_qblk, n_query, rm, index_base, d_score_bits, d_key);
        return hipGetLastError();
    }
    const float* tri = sliding_tri_table();
    if (!tri) return hipErrorOutOfMemory;
    // 64 K records per wave and chunk; a short overlap relative to the chunk keeps the repeated work small
    const uint32_t K = look <= 6u ? 1u : 4u;
    const uint32_t step = 64u * K - look;
    const uint64_t span = 64ull * K;
    const uint64_t n_chunks = n_pos <= span ? 1u : (n_pos - span + step - 1u) / step + 1u;
    const uint64_t want = (n_chunks + (kSlThreads / 64) - 1) / (kSlThreads / 64);
    const uint64_t cap = (uint64_t)cus * 6u;                             // 21 KB of LDS per workgroup
    const uint32_t grid = (uint32_t)(want < cap ? want : cap);
    if (K == 1)
        hipLaunchKernelGGL(compare_sliding_kernel<1>, dim3(grid), dim3(kSlThreads), 0, stream, d_recs, n_pos, d_qblk,
                           n_query, step, n_chunks, rm, tri, index_base, d_score_bits, d_key);
    else
        hipLaunchKernelGGL(compare_sliding_kernel<4>, dim3(grid), dim3(kSlThreads), 0, stream, d_recs, n_pos, d_qblk,
                           n_query, step, n_chunks, rm, tri, index_base, d_score_bits, d_key);
    return hipGetLastError();
}

}  // namespace lbad
