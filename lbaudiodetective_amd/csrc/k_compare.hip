// k_compare.hip -- fingerprint-vs-fingerprint compare and corpus top-1.
//
// Replaces LBAudioDetectiveFingerprintCompareSubfingerprints / ...CompareToFingerprint
// (LBAudioDetectiveFingerprint.m:119-176) and the best-match loop of the test harness
// (LBAudioDetectiveTests.m:57-91).
//
// Bit form of the per-sub-fingerprint rule.  A sub-fingerprint is a stream of Boolean pairs
// (pos, neg) at even bit positions.  With A the longer side's sub-fingerprint and B the other:
//   NZ       = (A | A >> 1) & EVEN & RANGE          pairs where A is non-zero      (Fp.m:159)
//   Y        = (A ^ B) | (A ^ B) >> 1               pairs that differ in either Boolean
//   possible = popc(NZ),  hits = popc(NZ & ~Y)      (Fp.m:160-167)
//   ratio    = hits / possible in float32, 0 when possible == 0   (Fp.m:171-175)
// Sums run in sub-fingerprint order in float32, then / (Float32)n2, then the running MAX over
// offsets (Fp.m:136-146), exactly as the oracle does.
//
// Two layouts:
//   slots  : entries[e][s][8 words]            (host fingerprints; one-off compares)
//   planes : tight bitstream of n_sub * Lp bits per entry (Lp = length rounded up to even),
//            cut into 16-byte planes stored plane-major so that lane e reads plane p at
//            planes[p * stride + e] -- fully coalesced 1 KiB per wave-instruction.  At the
//            default shape (5 x 200 Booleans) an entry is 1000 bits = 8 planes = 128 bytes of
//            HBM traffic for 125 bytes of information.
#include "internal.hpp"

#include <cmath>

#include <cstring>

namespace lbad {
namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ unsigned long long make_key(float score, uint64_t global_index) {
    return ((unsigned long long)__float_as_uint(score) << 32) |
           (unsigned long long)(0xFFFFFFFFu - (uint32_t)global_index);
}

template <int WAVES = kThreads / 64>
__device__ __forceinline__ void block_max_key(unsigned long long k, unsigned long long* out) {
    __shared__ unsigned long long s_k[WAVES];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(k, off, 64);
        k = o > k ? o : k;
    }
    if ((threadIdx.x & 63) == 0) s_k[threadIdx.x >> 6] = k;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long m = s_k[0];
        for (int i = 1; i < WAVES; ++i) m = s_k[i] > m ? s_k[i] : m;
        if (m) atomicMax(out, m);
    }
}

// Optional tail of a scan for the host-synchronous query.  Every workgroup stores its maximum in its own slot
// and takes a ticket (one contended atomic per workgroup instead of two: with 2048 workgroups the atomicMax on
// the single key word and the ticket cost ~25 us each, more than the scan of 1 M entries); the LAST workgroup
// to arrive reduces the slots, hands the key to the host through a pinned, host-coherent word pair
// {key, sequence number} and rearms the ticket.  The host polls the sequence number instead of paying a
// memset, a stream synchronisation and an 8-byte copy.
struct ScanFinish {
    unsigned int* ticket;                   // device, zero between queries
    unsigned long long* block_keys;         // device, one slot per workgroup
    volatile unsigned long long* host_out;  // pinned host memory mapped into the device: [0] key, [1] sequence
    unsigned long long seq;
};

template <int WAVES = kThreads / 64>
__device__ __forceinline__ unsigned long long block_reduce_max(unsigned long long k) {
    __shared__ unsigned long long s_m[WAVES];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(k, off, 64);
        k = o > k ? o : k;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = k;
    __syncthreads();
    unsigned long long m = s_m[0];
#pragma unroll
    for (int i = 1; i < WAVES; ++i) m = s_m[i] > m ? s_m[i] : m;
    return m;
}

template <int WAVES = kThreads / 64>
__device__ __forceinline__ void block_max_key_finish(unsigned long long k, unsigned long long* out, const ScanFinish fin) {
    if (fin.ticket == nullptr) {
        block_max_key<WAVES>(k, out);
        return;
    }
    __shared__ unsigned int s_last;
    const unsigned long long m = block_reduce_max<WAVES>(k);
    if (threadIdx.x == 0) {
        __hip_atomic_store(&fin.block_keys[blockIdx.x], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();                                    // the slot before the ticket
        s_last = atomicAdd(fin.ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (s_last) {
        __threadfence();
        unsigned long long best = 0ull;
        for (uint32_t b = threadIdx.x; b < gridDim.x; b += 64 * WAVES) {
            const unsigned long long v = __hip_atomic_load(&fin.block_keys[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            best = v > best ? v : best;
        }
        best = block_reduce_max<WAVES>(best);
        if (threadIdx.x == 0) {
            fin.host_out[0] = best;
            __threadfence_system();
            fin.host_out[1] = fin.seq;
            atomicExch(fin.ticket, 0u);
        }
    }
}

// even-position bits of word w (bit positions 32w .. 32w+31 of the sub-fingerprint) that are
// below 2 * pairs
__device__ __forceinline__ uint32_t range_mask(uint32_t w, uint32_t pairs) {
    const uint32_t lim = 2u * pairs;
    const uint32_t base = 32u * w;
    if (lim <= base) return 0u;
    const uint32_t nb = lim - base;
    const uint32_t m = nb >= 32u ? 0xFFFFFFFFu : ((1u << nb) - 1u);
    return m & 0x55555555u;
}

struct SubWords {
    uint32_t w[kPackedWords];
};

// word j of sub-fingerprint s of entry e, both layouts
template <bool PLANES>
__device__ __forceinline__ SubWords load_sub(const uint32_t* __restrict__ base, uint64_t e, uint64_t stride,
                                             uint32_t n_sub, uint32_t s, uint32_t lp) {
    SubWords r;
    if (!PLANES) {
        const uint4* p = reinterpret_cast<const uint4*>(base + (e * n_sub + s) * kPackedWords);
        const uint4 a = p[0], b = p[1];
        r.w[0] = a.x; r.w[1] = a.y; r.w[2] = a.z; r.w[3] = a.w;
        r.w[4] = b.x; r.w[5] = b.y; r.w[6] = b.z; r.w[7] = b.w;
    } else {
        const uint32_t total = (n_sub * lp + 31u) >> 5;  // words in the entry's stream (before plane padding)
        const uint32_t off = s * lp;
        const uint32_t w0 = off >> 5, sh = off & 31u;
        auto word = [&](uint32_t w) -> uint32_t {
            if (w >= total) return 0u;
            return base[((uint64_t)(w >> 2) * stride + e) * 4u + (w & 3u)];
        };
        const uint32_t nwords = (lp + 31u) >> 5;
        uint32_t prev = word(w0);
#pragma unroll
        for (uint32_t j = 0; j < kPackedWords; ++j) {
            uint32_t v = 0u;
            if (j < nwords) {
                const uint32_t next = word(w0 + j + 1);
                v = sh ? ((prev >> sh) | (next << (32u - sh))) : prev;
                prev = next;
                const uint32_t remaining = lp - 32u * j;
                if (remaining < 32u) v &= (1u << remaining) - 1u;
            }
            r.w[j] = v;
        }
    }
    return r;
}

__device__ __forceinline__ float sub_ratio(const SubWords& a, const SubWords& b, const uint32_t* s_mask) {
    uint32_t possible = 0, hits = 0;
#pragma unroll
    for (uint32_t w = 0; w < kPackedWords; ++w) {
        const uint32_t nz = (a.w[w] | (a.w[w] >> 1)) & s_mask[w];
        const uint32_t x = a.w[w] ^ b.w[w];
        const uint32_t y = x | (x >> 1);
        possible += __popc(nz);
        hits += __popc(nz & ~y);
    }
    return possible ? __fdiv_rn((float)hits, (float)possible) : 0.0f;
}

// One fingerprint against one fingerprint (LBAudioDetectiveFingerprintCompareToFingerprint, Fp.m:119-149), both
// in the slot layout and in global memory, any lengths: one thread per sliding offset walks the shorter side in
// sub-fingerprint order (the float32 sum of Fp.m:139-142 is sequential by definition, the offsets are not),
// the running MAX of Fp.m:144 becomes a block maximum and one atomicMax on the score's bit pattern (scores are
// >= 0, so their bits order like the values).  `a` is the side with at least as many sub-fingerprints.
__global__ __launch_bounds__(kThreads) void compare_pair_kernel(const uint32_t* __restrict__ a, uint32_t n1,
                                                                const uint32_t* __restrict__ b, uint32_t n2,
                                                                uint32_t pairs, unsigned int* __restrict__ out_bits) {
    __shared__ uint32_t s_mask[kPackedWords];
    __shared__ unsigned int s_best[kThreads / 64];
    if (threadIdx.x < kPackedWords) s_mask[threadIdx.x] = range_mask(threadIdx.x, pairs);
    __syncthreads();
    const uint32_t offsets = n1 - n2 + 1;
    unsigned int best = 0u;
    for (uint32_t o = blockIdx.x * kThreads + threadIdx.x; o < offsets; o += gridDim.x * kThreads) {
        float sum = 0.0f;
        for (uint32_t i = 0; i < n2; ++i) {
            const SubWords x = load_sub<false>(a, 0, 0, n1, i + o, 0);
            const SubWords y = load_sub<false>(b, 0, 0, n2, i, 0);
            sum = __fadd_rn(sum, sub_ratio(x, y, s_mask));
        }
        const unsigned int bits = __float_as_uint(__fdiv_rn(sum, (float)n2));
        best = bits > best ? bits : best;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned int v = __shfl_xor(best, off, 64);
        best = v > best ? v : best;
    }
    if ((threadIdx.x & 63) == 0) s_best[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int m = s_best[0];
        for (int i = 1; i < kThreads / 64; ++i) m = s_best[i] > m ? s_best[i] : m;
        if (m) atomicMax(out_bits, m);
    }
}

template <bool PLANES>
__global__ __launch_bounds__(kThreads) void compare_generic_kernel(
    const uint32_t* __restrict__ entries, uint64_t stride, uint64_t n_entries, uint32_t n_sub, uint32_t lp,
    const uint32_t* __restrict__ query, uint32_t n_query, uint32_t pairs, uint64_t index_base,
    float* __restrict__ scores, unsigned long long* __restrict__ key_out) {
    extern __shared__ uint32_t s_q[];  // n_query * 8 words
    __shared__ uint32_t s_mask[kPackedWords];
    for (uint32_t i = threadIdx.x; i < n_query * kPackedWords; i += kThreads) s_q[i] = query[i];
    if (threadIdx.x < kPackedWords) s_mask[threadIdx.x] = range_mask(threadIdx.x, pairs);
    __syncthreads();

    unsigned long long best = 0ull;
    for (uint64_t e = (uint64_t)blockIdx.x * kThreads + threadIdx.x; e < n_entries;
         e += (uint64_t)gridDim.x * kThreads) {
        // fp1 = query, fp2 = entry; the reference swaps so that "1" is the longer (Fp.m:123-131)
        const bool query_is_long = n_query >= n_sub;
        const uint32_t n1 = query_is_long ? n_query : n_sub;
        const uint32_t n2 = query_is_long ? n_sub : n_query;
        float match = 0.0f;
        for (uint32_t offset = 0; offset + n2 <= n1; ++offset) {
            float sum = 0.0f;
            for (uint32_t i = 0; i < n2; ++i) {
                SubWords a, b;
                if (query_is_long) {
#pragma unroll
                    for (uint32_t w = 0; w < kPackedWords; ++w) a.w[w] = s_q[(i + offset) * kPackedWords + w];
                    b = load_sub<PLANES>(entries, e, stride, n_sub, i, lp);
                } else {
                    a = load_sub<PLANES>(entries, e, stride, n_sub, i + offset, lp);
#pragma unroll
                    for (uint32_t w = 0; w < kPackedWords; ++w) b.w[w] = s_q[i * kPackedWords + w];
                }
                sum = __fadd_rn(sum, sub_ratio(a, b, s_mask));
            }
            const float cand = __fdiv_rn(sum, (float)n2);
            match = (match < cand) ? cand : match;  // Foundation MAX(A,B) = a < b ? b : a
        }
        if (scores) scores[e] = match;
        const unsigned long long k = make_key(match, index_base + e);
        best = k > best ? k : best;
    }
    block_max_key(best, key_out);
}

// ---- slots -> planes ----------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void pack_planes_kernel(const uint32_t* __restrict__ slots,
                                                               uint64_t n_entries, uint32_t n_sub, uint32_t lp,
                                                               uint32_t n_planes, uint4* __restrict__ planes,
                                                               uint64_t stride, uint64_t first) {
    const uint64_t e = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    const uint32_t p = blockIdx.y;
    if (e >= n_entries || p >= n_planes) return;
    const uint32_t total_bits = n_sub * lp;
    const uint32_t* ent = slots + e * n_sub * kPackedWords;
    uint32_t out[4];
#pragma unroll
    for (uint32_t q = 0; q < 4; ++q) {
        uint32_t word = 0u, filled = 0u;
        uint32_t pos = (p * 4u + q) * 32u;
        while (filled < 32u && pos < total_bits) {
            const uint32_t s = pos / lp, o = pos % lp;
            uint32_t take = 32u - filled;
            if (take > lp - o) take = lp - o;
            const uint32_t* sw = ent + s * kPackedWords;
            const uint32_t wi = o >> 5, sh = o & 31u;
            uint64_t win = sw[wi];
            if (wi + 1 < kPackedWords) win |= (uint64_t)sw[wi + 1] << 32;
            uint32_t bits = (uint32_t)(win >> sh);
            if (take < 32u) bits &= (1u << take) - 1u;
            word |= bits << filled;
            filled += take;
            pos += take;
        }
        out[q] = word;
    }
    planes[(uint64_t)p * stride + first + e] = make_uint4(out[0], out[1], out[2], out[3]);
}

// ---- specialised scan: Lp = 200, n_query == n_sub == NSUB, query supplies the mask --------
constexpr uint32_t kLp = 200;
constexpr uint32_t kSubSpan = 7;  // a 200-bit field at an even multiple-of-8 bit offset touches <= 7 words

constexpr uint32_t kPlaneQueryWords = 144;   // room for NSUB = 8: 13 planes x 4 + 8 x kSubSpan masks + 8 possible + 8 + 8 reciprocal pairs = 132

template <int NSUB>
struct PlaneShape {
    static constexpr uint32_t bits = NSUB * kLp;
    static constexpr uint32_t words = (bits + 31) / 32;
    static constexpr uint32_t planes = (bits + 127) / 128;
    // constant block handed to the kernel: query stream, per-(sub, span word) NZ masks, possible counts
    static constexpr uint32_t off_mask = planes * 4;
    static constexpr uint32_t off_possible = off_mask + NSUB * kSubSpan;
    // round 6 (the batch scan): rh = RN(1 / possible), rl = RN(fma(-possible, rh, 1) * rh) per sub-fingerprint, 0 0 where nothing is possible
    static constexpr uint32_t off_rh = off_possible + NSUB;
    static constexpr uint32_t off_rl = off_rh + NSUB;
    static constexpr uint32_t total = off_rl + NSUB;
};

// The query block travels in the kernel-argument segment: every lane reads the same words, so the
// compiler keeps them in SGPRs (scalar loads) -- no staging copy, no LDS reads in the scan loop.
struct PlaneQueryArg {
    uint32_t w[kPlaneQueryWords];
};

// The single-query scan runs ONE workgroup of 1024 threads per CU: with 256 threads a CU had four waves and eight 16-byte
// loads each in flight, and a million entries (15 trips per thread) took 26.3 us of mostly exposed latency; sixteen waves
// per CU: 23.3 us = 5.4 TB/s of the Infinity Cache (10 M entries: 217 us either way, tools/exp/query_latency.py).
#ifndef LBAD_PLANE_THREADS
#define LBAD_PLANE_THREADS 1024
#endif
#ifndef LBAD_PLANE_WG_SMALL
#define LBAD_PLANE_WG_SMALL 256
#endif
#ifndef LBAD_PLANE_WG_LARGE
#define LBAD_PLANE_WG_LARGE 256
#endif
constexpr int kPlaneThreads = LBAD_PLANE_THREADS;      // the single-query scan

template <int NSUB>
__global__ __launch_bounds__(kPlaneThreads) void compare_planes_kernel(const uint4* __restrict__ planes, uint64_t stride,
                                                                  uint64_t n_entries, const PlaneQueryArg qc,
                                                                  uint64_t index_base, float* __restrict__ scores,
                                                                  unsigned long long* __restrict__ key_out,
                                                                  const ScanFinish fin) {
    using S = PlaneShape<NSUB>;
    unsigned long long best = 0ull;
    for (uint64_t e = (uint64_t)blockIdx.x * kPlaneThreads + threadIdx.x; e < n_entries;
         e += (uint64_t)gridDim.x * kPlaneThreads) {
        uint32_t y[S::planes * 4];
#pragma unroll
        for (uint32_t p = 0; p < S::planes; ++p) {
            const uint4 v = planes[(uint64_t)p * stride + e];
            y[4 * p + 0] = v.x; y[4 * p + 1] = v.y; y[4 * p + 2] = v.z; y[4 * p + 3] = v.w;
        }
#pragma unroll
        for (uint32_t w = 0; w < S::planes * 4; ++w) {
            const uint32_t x = y[w] ^ qc.w[w];
            y[w] = ~(x | (x >> 1));
        }
        float sum = 0.0f;
#pragma unroll
        for (uint32_t s = 0; s < (uint32_t)NSUB; ++s) {
            const uint32_t w0 = (s * kLp) >> 5;
            uint32_t hits = 0;
#pragma unroll
            for (uint32_t j = 0; j < kSubSpan; ++j) {
                if (w0 + j < S::planes * 4) hits += __popc(y[w0 + j] & qc.w[S::off_mask + s * kSubSpan + j]);
            }
            const float possible = __uint_as_float(qc.w[S::off_possible + s]);
            const float ratio = possible > 0.0f ? __fdiv_rn((float)hits, possible) : 0.0f;
            sum = __fadd_rn(sum, ratio);
        }
        const float cand = __fdiv_rn(sum, (float)NSUB);
        const float match = (0.0f < cand) ? cand : 0.0f;
        if (scores) scores[e] = match;
        const unsigned long long k = make_key(match, index_base + e);
        best = k > best ? k : best;
    }
    block_max_key_finish<kPlaneThreads / 64>(best, key_out, fin);
}

// Batch form: up to QB queries share one pass over the corpus.  Query blocks (kPlaneQueryWords each) sit in global memory and
// are read with wave-uniform indices, i.e. through the scalar cache.
// Round 6: eight queries took 0.65 ms against 10 M entries (3.0 x one query; tools/exp/uniform_batch_time.py) -- not HBM-bound
// as the round-2 comment here assumed but 280 vector instructions per (entry, query).  Now per 32-bit word of a sub-fingerprint's
// span TWO three-input operations and a count: u = mask & ~(b ^ q), t = u & ~((b >> 1) ^ (q >> 1)) -- at an even bit the first
// is "first Booleans equal", the second "second Booleans equal"; b >> 1 is taken once per entry for all queries, q >> 1 is a
// scalar instruction -- and the quotient by two fused multiply-adds on (rh, rl) from the query block instead of an IEEE division
// per sub-fingerprint: 125 instructions per (entry, query).
constexpr int kQueryBatch = 8;

template <int NSUB>
__global__ __launch_bounds__(kThreads) void compare_planes_batch_kernel(const uint4* __restrict__ planes,
                                                                        uint64_t stride, uint64_t n_entries,
                                                                        const uint32_t* __restrict__ qblocks,
                                                                        uint32_t n_queries, uint64_t index_base,
                                                                        unsigned long long* __restrict__ keys_out) {
    using S = PlaneShape<NSUB>;
    unsigned long long best[kQueryBatch];
#pragma unroll
    for (int q = 0; q < kQueryBatch; ++q) best[q] = 0ull;
    for (uint64_t e = (uint64_t)blockIdx.x * kThreads + threadIdx.x; e < n_entries;
         e += (uint64_t)gridDim.x * kThreads) {
        uint32_t b[S::planes * 4], b1[S::planes * 4];
#pragma unroll
        for (uint32_t p = 0; p < S::planes; ++p) {
            const uint4 v = planes[(uint64_t)p * stride + e];
            b[4 * p + 0] = v.x; b[4 * p + 1] = v.y; b[4 * p + 2] = v.z; b[4 * p + 3] = v.w;
        }
#pragma unroll
        for (uint32_t w = 0; w < S::planes * 4; ++w) b1[w] = b[w] >> 1;       // (pairs never straddle a word: bit 2 p + 1 lands on 2 p)
#pragma unroll
        for (int q = 0; q < kQueryBatch; ++q) {
            if ((uint32_t)q < n_queries) {
                const uint32_t* qc = qblocks + (size_t)q * kPlaneQueryWords;
                float sum = 0.0f;
#pragma unroll
                for (uint32_t s = 0; s < (uint32_t)NSUB; ++s) {
                    const uint32_t w0 = (s * kLp) >> 5;
                    uint32_t h = 0x4B000000u;                      // hits counted on top of the bits of 2^23
#pragma unroll
                    for (uint32_t j = 0; j < kSubSpan; ++j) {
                        if (w0 + j < S::planes * 4) {
                            const uint32_t qw = qc[w0 + j];
                            const uint32_t u = __builtin_amdgcn_bitop3_b32(qc[S::off_mask + s * kSubSpan + j], b[w0 + j], qw, 0x90);   // a & ~(b ^ c)
                            const uint32_t t = __builtin_amdgcn_bitop3_b32(u, b1[w0 + j], qw >> 1, 0x90);
                            h += __popc(t);
                        }
                    }
                    const float hf = __fsub_rn(__uint_as_float(h), 8388608.0f);
                    const float rh = __uint_as_float(qc[S::off_rh + s]), rl = __uint_as_float(qc[S::off_rl + s]);
                    sum = __fadd_rn(sum, __fmaf_rn(hf, rh, __fmul_rn(hf, rl)));      // == hits / possible, 0 where nothing is possible
                }
                const float cand = __fdiv_rn(sum, (float)NSUB);
                const float match = (0.0f < cand) ? cand : 0.0f;
                const unsigned long long k = make_key(match, index_base + e);
                best[q] = k > best[q] ? k : best[q];
            }
        }
    }
#pragma unroll
    for (int q = 0; q < kQueryBatch; ++q) {
        if ((uint32_t)q < n_queries) block_max_key(best[q], keys_out + q);
        __syncthreads();   // block_max_key reuses one LDS scratch array
    }
}

uint32_t grid_for(uint64_t n_entries) {
    const uint64_t blocks = (n_entries + kThreads - 1) / kThreads;
    // One or two workgroups per CU, grid-stride the rest.  Every workgroup ends in an atomic on ONE word (the key,
    // or the ticket of the polled query); at 8 per CU those 2048 atomics took longer than the scan itself
    // (1 M entries: 36.4 us at 2048 workgroups, 25.9 at 256; 10 M: 228.8 / 215.4 us at 2048 / 512, 257.6 at 256).
    const uint64_t cap = n_entries <= (1ull << 21) ? 256ull : 512ull;
    return (uint32_t)(blocks < cap ? (blocks ? blocks : 1) : cap);
}

template <int NSUB>
hipError_t launch_planes_n(const uint4* d_planes, uint64_t stride, uint64_t n_entries, const uint32_t* h_qc,
                           uint64_t index_base, float* d_scores, unsigned long long* d_key, hipStream_t stream,
                           const ScanFinish& fin) {
    static_assert(PlaneShape<NSUB>::total <= kPlaneQueryWords, "query block does not fit the kernel argument");
    PlaneQueryArg arg;
    std::memset(&arg, 0, sizeof(arg));
    std::memcpy(arg.w, h_qc, PlaneShape<NSUB>::total * sizeof(uint32_t));
    const uint64_t blocks = (n_entries + kPlaneThreads - 1) / kPlaneThreads;
    const uint64_t cap = n_entries <= (1ull << 21) ? (uint64_t)LBAD_PLANE_WG_SMALL : (uint64_t)LBAD_PLANE_WG_LARGE;
    const uint32_t grid = (uint32_t)(blocks < cap ? (blocks ? blocks : 1) : cap);
    hipLaunchKernelGGL(compare_planes_kernel<NSUB>, dim3(grid), dim3(kPlaneThreads), 0, stream, d_planes,
                       stride, n_entries, arg, index_base, d_scores, d_key, fin);
    return hipGetLastError();
}

}  // namespace

uint32_t planes_per_entry(uint32_t subfp_len, uint32_t n_sub) {
    const uint32_t lp = subfp_len + (subfp_len & 1u);
    return (n_sub * lp + 127u) / 128u;
}

bool planes_supported(uint32_t subfp_len, uint32_t n_sub) {
    // the tight layout itself is generic; cap the entry at 64 planes (8192 bits)
    return subfp_len >= 2 && subfp_len <= LBAD_MAX_SUBFINGERPRINT_LENGTH && n_sub >= 1 &&
           planes_per_entry(subfp_len, n_sub) <= 64;
}

hipError_t launch_pack_planes(const uint32_t* d_slots, uint64_t n_entries, uint32_t n_sub, uint32_t subfp_len,
                              uint4* d_planes, uint64_t plane_stride, uint64_t first, hipStream_t stream) {
    if (n_entries == 0) return hipSuccess;
    const uint32_t lp = subfp_len + (subfp_len & 1u);
    const uint32_t np = planes_per_entry(subfp_len, n_sub);
    const uint64_t bx = (n_entries + kThreads - 1) / kThreads;
    if (bx > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pack_planes_kernel, dim3((uint32_t)bx, np), dim3(kThreads), 0, stream, d_slots, n_entries,
                       n_sub, lp, np, d_planes, plane_stride, first);
    return hipGetLastError();
}

hipError_t launch_compare_slots(const uint32_t* d_entries, uint64_t n_entries, uint32_t n_sub, uint32_t subfp_len,
                                const uint32_t* d_query, uint32_t n_query, uint32_t range, uint64_t index_base,
                                float* d_scores, unsigned long long* d_key, hipStream_t stream) {
    if (n_entries == 0) return hipSuccess;
    const uint32_t lim = range < subfp_len ? range : subfp_len;
    const uint32_t pairs = (lim + 1u) / 2u;
    const uint32_t lp = subfp_len + (subfp_len & 1u);
    const size_t lds = (size_t)n_query * kPackedWords * sizeof(uint32_t);
    hipLaunchKernelGGL(compare_generic_kernel<false>, dim3(grid_for(n_entries)), dim3(kThreads), lds, stream,
                       d_entries, (uint64_t)0, n_entries, n_sub, lp, d_query, n_query, pairs, index_base, d_scores,
                       d_key);
    return hipGetLastError();
}

hipError_t launch_compare_planes_generic(const uint4* d_planes, uint64_t plane_stride, uint64_t n_entries,
                                         uint32_t n_sub, uint32_t subfp_len, const uint32_t* d_query,
                                         uint32_t n_query, uint32_t range, uint64_t index_base, float* d_scores,
                                         unsigned long long* d_key, hipStream_t stream) {
    if (n_entries == 0) return hipSuccess;
    const uint32_t lim = range < subfp_len ? range : subfp_len;
    const uint32_t pairs = (lim + 1u) / 2u;
    const uint32_t lp = subfp_len + (subfp_len & 1u);
    const size_t lds = (size_t)n_query * kPackedWords * sizeof(uint32_t);
    hipLaunchKernelGGL(compare_generic_kernel<true>, dim3(grid_for(n_entries)), dim3(kThreads), lds, stream,
                       reinterpret_cast<const uint32_t*>(d_planes), plane_stride, n_entries, n_sub, lp, d_query,
                       n_query, pairs, index_base, d_scores, d_key);
    return hipGetLastError();
}

bool planes_fast_supported(uint32_t subfp_len, uint32_t n_sub, uint32_t n_query) {
    return subfp_len == kLp && n_query == n_sub && n_sub >= 1 && n_sub <= 8;
}

uint32_t planes_fast_const_words(uint32_t n_sub) {
    const uint32_t planes = (n_sub * kLp + 127) / 128;
    return planes * 4 + n_sub * kSubSpan + 3 * n_sub;
}

// Host: build the constant block of the specialised kernel from the query's slot words.
void build_plane_query(const uint32_t* q_slots, uint32_t n_sub, uint32_t range, std::vector<uint32_t>& out) {
    const uint32_t planes = (n_sub * kLp + 127) / 128;
    const uint32_t words = planes * 4;
    out.assign(planes_fast_const_words(n_sub), 0u);
    const uint32_t lim = range < kLp ? range : kLp;
    const uint32_t pairs = (lim + 1u) / 2u;
    for (uint32_t s = 0; s < n_sub; ++s) {
        uint32_t possible = 0;
        for (uint32_t b = 0; b < kLp; ++b) {
            const uint32_t bit = (q_slots[s * kPackedWords + (b >> 5)] >> (b & 31)) & 1u;
            const uint32_t pos = s * kLp + b;
            if (bit) out[pos >> 5] |= 1u << (pos & 31);
        }
        const uint32_t w0 = (s * kLp) >> 5;
        for (uint32_t p = 0; p < pairs; ++p) {
            const uint32_t b0 = 2 * p, b1 = 2 * p + 1;
            const uint32_t v0 = (q_slots[s * kPackedWords + (b0 >> 5)] >> (b0 & 31)) & 1u;
            const uint32_t v1 = b1 < kLp ? (q_slots[s * kPackedWords + (b1 >> 5)] >> (b1 & 31)) & 1u : 0u;
            if (v0 | v1) {
                ++possible;
                const uint32_t pos = s * kLp + b0;
                const uint32_t j = (pos >> 5) - w0;
                out[words + s * kSubSpan + j] |= 1u << (pos & 31);
            }
        }
        const float pf = (float)possible;
        uint32_t pbits;
        memcpy(&pbits, &pf, 4);
        out[words + n_sub * kSubSpan + s] = pbits;
        // the quotient hits / possible by two fused multiply-adds (tools/verify_ratio_fma.c: exact for every hits <= possible)
        const float rh = possible ? 1.0f / pf : 0.0f;
        const float rl = std::fmaf(-pf, rh, possible ? 1.0f : 0.0f) * rh;
        memcpy(&out[words + n_sub * kSubSpan + n_sub + s], &rh, 4);
        memcpy(&out[words + n_sub * kSubSpan + 2 * n_sub + s], &rl, 4);
    }
}

// h_qc: HOST pointer to the block built by build_plane_query (it is passed as a kernel argument)
hipError_t launch_compare_planes_fast(const uint4* d_planes, uint64_t plane_stride, uint64_t n_entries,
                                      uint32_t n_sub, const uint32_t* d_qc, uint64_t index_base, float* d_scores,
                                      unsigned long long* d_key, hipStream_t stream, unsigned int* d_ticket,
                                      unsigned long long* host_out_dev, unsigned long long seq) {
    if (n_entries == 0) return hipSuccess;
    ScanFinish fin;
    fin.ticket = d_ticket;
    fin.block_keys = d_key;                 // polled mode: d_key is the per-workgroup slot array (kScanSlots words)
    fin.host_out = host_out_dev;
    fin.seq = seq;
    switch (n_sub) {
        case 1: return launch_planes_n<1>(d_planes, plane_stride, n_entries, d_qc, index_base, d_scores, d_key, stream, fin);
        case 2: return launch_planes_n<2>(d_planes, plane_stride, n_entries, d_qc, index_base, d_scores, d_key, stream, fin);
        case 3: return launch_planes_n<3>(d_planes, plane_stride, n_entries, d_qc, index_base, d_scores, d_key, stream, fin);
        case 4: return launch_planes_n<4>(d_planes, plane_stride, n_entries, d_qc, index_base, d_scores, d_key, stream, fin);
        case 5: return launch_planes_n<5>(d_planes, plane_stride, n_entries, d_qc, index_base, d_scores, d_key, stream, fin);
        case 6: return launch_planes_n<6>(d_planes, plane_stride, n_entries, d_qc, index_base, d_scores, d_key, stream, fin);
        case 7: return launch_planes_n<7>(d_planes, plane_stride, n_entries, d_qc, index_base, d_scores, d_key, stream, fin);
        case 8: return launch_planes_n<8>(d_planes, plane_stride, n_entries, d_qc, index_base, d_scores, d_key, stream, fin);
        default: return hipErrorNotSupported;
    }
}


template <int NSUB>
static hipError_t launch_batch_n(const uint4* d_planes, uint64_t stride, uint64_t n_entries, const uint32_t* d_qblocks,
                                 uint32_t n_queries, uint64_t index_base, unsigned long long* d_keys,
                                 hipStream_t stream) {
    for (uint32_t q0 = 0; q0 < n_queries; q0 += kQueryBatch) {
        const uint32_t nq = n_queries - q0 < (uint32_t)kQueryBatch ? n_queries - q0 : (uint32_t)kQueryBatch;
        hipLaunchKernelGGL(compare_planes_batch_kernel<NSUB>, dim3(grid_for(n_entries)), dim3(kThreads), 0, stream,
                           d_planes, stride, n_entries, d_qblocks + (size_t)q0 * kPlaneQueryWords, nq, index_base,
                           d_keys + q0);
    }
    return hipGetLastError();
}

uint32_t plane_query_words() { return kPlaneQueryWords; }

// d_qblocks: n_queries blocks of plane_query_words() words each (build_plane_query output, zero padded)
hipError_t launch_compare_planes_batch(const uint4* d_planes, uint64_t plane_stride, uint64_t n_entries, uint32_t n_sub,
                                       const uint32_t* d_qblocks, uint32_t n_queries, uint64_t index_base,
                                       unsigned long long* d_keys, hipStream_t stream) {
    if (n_entries == 0 || n_queries == 0) return hipSuccess;
    switch (n_sub) {
        case 1: return launch_batch_n<1>(d_planes, plane_stride, n_entries, d_qblocks, n_queries, index_base, d_keys, stream);
        case 2: return launch_batch_n<2>(d_planes, plane_stride, n_entries, d_qblocks, n_queries, index_base, d_keys, stream);
        case 3: return launch_batch_n<3>(d_planes, plane_stride, n_entries, d_qblocks, n_queries, index_base, d_keys, stream);
        case 4: return launch_batch_n<4>(d_planes, plane_stride, n_entries, d_qblocks, n_queries, index_base, d_keys, stream);
        case 5: return launch_batch_n<5>(d_planes, plane_stride, n_entries, d_qblocks, n_queries, index_base, d_keys, stream);
        case 6: return launch_batch_n<6>(d_planes, plane_stride, n_entries, d_qblocks, n_queries, index_base, d_keys, stream);
        case 7: return launch_batch_n<7>(d_planes, plane_stride, n_entries, d_qblocks, n_queries, index_base, d_keys, stream);
        case 8: return launch_batch_n<8>(d_planes, plane_stride, n_entries, d_qblocks, n_queries, index_base, d_keys, stream);
        default: return hipErrorNotSupported;
    }
}

hipError_t launch_compare_pair(const uint32_t* d_a, uint32_t n1, const uint32_t* d_b, uint32_t n2, uint32_t subfp_len,
                               uint32_t range, unsigned int* d_out_bits, hipStream_t stream) {
    if (n1 < n2 || n2 == 0) return hipErrorInvalidValue;
    const uint32_t lim = range < subfp_len ? range : subfp_len;   // Fp.m:155
    const uint32_t pairs = (lim + 1) / 2;
    const uint32_t offsets = n1 - n2 + 1;
    uint32_t grid = (offsets + kThreads - 1) / kThreads;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(compare_pair_kernel, dim3(grid), dim3(kThreads), 0, stream, d_a, n1, d_b, n2, pairs, d_out_bits);
    return hipGetLastError();
}

}  // namespace lbad
