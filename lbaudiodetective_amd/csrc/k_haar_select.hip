// k_haar_select.hip -- unfused stage 2: 128 x bands frame -> packed sub-fingerprint.
//
// Replaces, per full frame, LBAudioDetectiveFrameDecompose (LBAudioDetectiveFrame.m:113-153),
// LBAudioDetectiveFrameExtractFingerprint (LBAudioDetectiveFrame.m:165-191) and the truncating
// copy of LBAudioDetectiveFingerprintAddSubfingerprint (LBAudioDetectiveFingerprint.m:91-100,
// called from LBAudioDetective.m:326-328).
//
// The reference sorts all 128*bands coefficients and keeps the signs of the first few.  Here
// one workgroup per frame (a) runs the 2-D Haar in LDS, (b) bisects the integer |v| key space
// for a threshold that leaves between `keep` and 256 candidates, (c) ranks only those
// candidates against each other (ties: lower flat index first, like the oracle) and (d) sets
// the sign bits of ranks < keep.  The result is identical to a full stable sort.
#include "internal.hpp"

#include <cmath>

namespace lbad {
namespace {

constexpr int kThreads = 256;
constexpr uint32_t kCand = 128;      // candidates ranked exhaustively (keep <= 128: sub-fingerprints of <= 256 Booleans)

// 1-D Haar of LBAudioDetectiveFrameDecomposeArray applied to `lines` independent lines of length `len`; element e
// of line l sits at [l * lstride + e * estride] of every buffer.  The reference works in place through a scratch
// copy (Frame.m:133-153); here the detail values of a level -- final once written -- go straight to `out`, and
// the sums the next level consumes alternate between `scratch` and the part of `in` the previous level has
// finished reading: one barrier and one store per value and level.  `in` is destroyed.
// `root` = sqrtf(len) is computed on the host: the device's sqrt is the 1-ulp v_sqrt_f32 (sqrtf(14.0f)
// comes out one ulp low, for example), whatever -fhip-fp32-correctly-rounded-divide-sqrt promises.
__device__ void haar_lines(float* in, float* out, float* scratch, uint32_t lines, uint32_t len, uint32_t lstride,
                           uint32_t estride, float root) {
    const float root2 = __fsqrt_rn(2.0f);   // constant-folded by the compiler (correctly rounded)
    if (len == 1) {
        for (uint32_t l = threadIdx.x; l < lines; l += kThreads) out[l * lstride] = __fdiv_rn(in[l * lstride], root);
        __syncthreads();
        return;
    }
    // `scratch` holds len / 2 values per line, packed
    struct View { float* p; uint32_t ls, es; };
    View src{in, lstride, estride}, dst{scratch, len >> 1, 1u};
    bool first = true;
    uint32_t have = len;                 // values of the running line: src[0 .. have)
    while (have > 1) {
        const uint32_t cnt = have >> 1;
        for (uint32_t p = threadIdx.x; p < lines * cnt; p += kThreads) {
            const uint32_t l = p / cnt, i = p % cnt;
            float ev = src.p[l * src.ls + (2 * i) * src.es];
            float od = src.p[l * src.ls + (2 * i + 1) * src.es];
            if (first) {                                                // the pre-scale of Frame.m:137-139
                ev = __fdiv_rn(ev, root);
                od = __fdiv_rn(od, root);
            }
            const float sum = __fdiv_rn(__fadd_rn(ev, od), root2);
            const float dif = __fdiv_rn(__fsub_rn(ev, od), root2);
            out[l * lstride + (cnt + i) * estride] = dif;
            if (cnt == 1) out[l * lstride] = sum;
            else dst.p[l * dst.ls + i * dst.es] = sum;
        }
        if (have & 1u) {
            // an odd count leaves its last value where it is for good (the reference copies 2 cnt values back)
            for (uint32_t l = threadIdx.x; l < lines; l += kThreads) {
                const float v = src.p[l * src.ls + (have - 1) * src.es];
                out[l * lstride + (have - 1) * estride] = first ? __fdiv_rn(v, root) : v;
            }
        }
        __syncthreads();
        // this level has read src[0 .. have): the next one may write its sums there
        const View t = src;
        src = dst;
        dst = t;
        first = false;
        have = cnt;
    }
}

constexpr uint32_t kHistBuckets = 2048;   // |v| bits >> 20

// inclusive prefix sum over the 64 lanes of a wave (full EXEC): four row shifts, two row broadcasts (gfx9 DPP)
__device__ __forceinline__ uint32_t wave_prefix_sum(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111 /* row_shr:1 */, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112 /* row_shr:2 */, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114 /* row_shr:4 */, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118 /* row_shr:8 */, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142 /* row_bcast:15 */, 0xA, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143 /* row_bcast:31 */, 0xC, 0xF, false);
    return x;
}

// PER = coefficients per thread (128 x bands / 256, rounded up to 8 / 16 / 32): the select works on registers
template <int PER>
__global__ __launch_bounds__(kThreads) void haar_select_kernel(const float* __restrict__ frames, uint32_t bands,
                                                               float root_bands, uint32_t keep, uint32_t subfp_len,
                                                               uint32_t* __restrict__ packed,
                                                               float* __restrict__ haar_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const uint32_t n = kRowsPerFrame * bands;
    float* a = smem;                   // the frame, then the finished coefficients
    float* b = smem + n;               // rows done, columns still to do
    float* tmp = smem + 2 * n;         // running sums (n / 2 floats, packed per line)
    __shared__ uint32_t s_red[8];
    __shared__ __attribute__((aligned(16))) unsigned long long s_cand[kCand];
    __shared__ uint32_t s_rank[kCand];
    __shared__ uint32_t s_ncand;
    __shared__ uint32_t s_bits[kPackedWords];

    const int t = threadIdx.x;
    const uint64_t frame = blockIdx.x;
    const float* src = frames + frame * n;
    for (uint32_t i = t; i < n; i += kThreads) a[i] = src[i];
    if (t < (int)kPackedWords) s_bits[t] = 0;
    if (t < (int)kCand) {
        s_rank[t] = 0;
        s_cand[t] = 0;                 // zero keys pad the list to a multiple of 8 for the ranking loop
    }
    if (t == 0) s_ncand = 0;
    __syncthreads();

    haar_lines(a, b, tmp, kRowsPerFrame, bands, bands, 1, root_bands);                          // every row (Frame.m:114-116)
    haar_lines(b, a, tmp, bands, kRowsPerFrame, 1, bands, __fsqrt_rn((float)kRowsPerFrame));    // every column (:118-131); folded

    if (haar_out) {
        float* dst = haar_out + frame * n;
        for (uint32_t i = t; i < n; i += kThreads) dst[i] = a[i];
    }

    // this thread's coefficients: flat index t + 256 j; a slot past the frame gets key 0 and is never selected
    float v[PER];
    uint32_t key[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const uint32_t idx = (uint32_t)t + 256u * j;
        v[j] = idx < n ? a[idx] : 0.0f;
        key[j] = __float_as_uint(v[j]) & 0x7fffffffu;
    }
    auto block_count = [&](auto pred, int parity) -> uint32_t {
        // per-wave counts are scalar popcounts of lane masks; one LDS word per wave, double-buffered
        uint32_t c = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) c += (uint32_t)__popcll(__ballot(pred(j)));
        uint32_t* slot = s_red + 4 * parity;
        if ((t & 63) == 0) slot[t >> 6] = c;
        __syncthreads();
        return slot[0] + slot[1] + slot[2] + slot[3];
    };

    // ---- threshold search on key = |v| bits ------------------------------------------------
    uint32_t lo = 0, hi = 0x80000000u;  // count(key >= lo) >= keep > count(key >= hi)
    uint32_t cnt_lo = n;
    uint32_t idx_bound = n;             // among key == lo only flat indices < idx_bound are candidates
    int parity = 0;
    // First by histogram, as in k_haar_select32.hip (round 4): 2048 counters over |v| bits >> 20 in the LDS the Haar has
    // left (the launcher reserves at least 8 KB), summed from the top by one wave; the bucket in which the count of keys
    // >= its lower bound first reaches `keep` gives the bracket [b, b + 1) << 20 -- final when at most kCand keys lie above
    // its lower end (the usual case), else the bisection below finishes inside it.  Bucket 0 is never counted.
    {
        uint32_t* hist = reinterpret_cast<uint32_t*>(smem);
        __shared__ uint32_t s_sel[4];
        __syncthreads();                                   // every thread holds its coefficients in registers
        for (int i = t; i < (int)(kHistBuckets / 4); i += kThreads) reinterpret_cast<uint4*>(hist)[i] = uint4{0u, 0u, 0u, 0u};
        __syncthreads();
        {
            uint32_t any = 0;
#pragma unroll
            for (int j = 0; j < PER; ++j) any |= key[j];
            if (any) {
#pragma unroll
                for (int j = 0; j < PER; ++j) atomicAdd(&hist[key[j] >> 20], 1u);     // (padding keys are 0: bucket 0)
            }
        }
        __syncthreads();
        if (t < 64) {
            const uint4* hsrc = reinterpret_cast<const uint4*>(hist) + (kHistBuckets / 4 - 8 - 8 * t);
            uint32_t own = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint4 c = hsrc[q];
                own += (c.x + c.y) + (c.z + c.w);
            }
            if (t == 63) own -= hist[0];
            const uint32_t incl = wave_prefix_sum(own), excl = incl - own;
            const unsigned long long m1 = __ballot(excl < keep && keep <= incl);
            uint32_t out_lo = 0, out_hi = 1u << 20, out_cnt = n;
            if (m1 != 0ull) {
                const int L = __ffsll((long long)m1) - 1;
                const uint32_t before = (uint32_t)__builtin_amdgcn_readlane((int)excl, L);
                const uint32_t top = kHistBuckets - 1u - 32u * (uint32_t)L;
                const uint32_t h = t < 32 ? hist[top - (uint32_t)t] : 0u;
                const uint32_t p = before + wave_prefix_sum(h);
                const unsigned long long m2 = __ballot(t < 32 && p - h < keep && keep <= p);
                const int J = __ffsll((long long)m2) - 1;
                const uint32_t bkt = top - (uint32_t)J;
                if (m2 != 0ull && bkt != 0u) {
                    out_lo = bkt << 20;
                    out_hi = (bkt + 1u) << 20;
                    out_cnt = (uint32_t)__builtin_amdgcn_readlane((int)p, J);
                }
            }
            if (t == 0) { s_sel[0] = out_lo; s_sel[1] = out_hi; s_sel[2] = out_cnt; }
        }
        __syncthreads();
        lo = s_sel[0];
        hi = s_sel[1];
        cnt_lo = s_sel[2];
    }
    while (cnt_lo > kCand && hi - lo > 1) {
        const uint32_t mid = lo + ((hi - lo) >> 1);     // >= 1: the padding keys never count
        const uint32_t c = block_count([&](int j) { return key[j] >= mid; }, parity);
        parity ^= 1;
        if (c >= keep) { lo = mid; cnt_lo = c; } else { hi = mid; }
    }
    if (cnt_lo > kCand) {
        // plateau: more than kCand coefficients share the threshold key `lo`.  Take every
        // key > lo, and of the equal ones the lowest flat indices until `keep` is reached.
        const uint32_t g = block_count([&](int j) { return key[j] > lo; }, parity);
        parity ^= 1;
        uint32_t ilo = 0, ihi = n;      // smallest bound with g + ties(idx < bound) >= keep
        while (ilo < ihi) {
            const uint32_t im = ilo + ((ihi - ilo) >> 1);
            const uint32_t c = block_count([&](int j) { return key[j] == lo && (uint32_t)t + 256u * j < im && (uint32_t)t + 256u * j < n; }, parity);
            parity ^= 1;
            if (g + c >= keep) ihi = im; else ilo = im + 1;
        }
        idx_bound = ilo;
    }

    // ---- gather candidates: composite = key << 15 | (8191 - idx) << 2 | sign code; one LDS atomic per wave ----
    {
        uint32_t wave_total = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const uint32_t idx = (uint32_t)t + 256u * j;
            const bool sel = idx < n && (key[j] > lo || (key[j] == lo && idx < idx_bound));
            wave_total += (uint32_t)__popcll(__ballot(sel));
        }
        uint32_t base = 0;
        if ((t & 63) == 0) base = atomicAdd(&s_ncand, wave_total);
        base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const uint32_t idx = (uint32_t)t + 256u * j;
            const bool sel = idx < n && (key[j] > lo || (key[j] == lo && idx < idx_bound));
            const unsigned long long m = __ballot(sel);
            if (sel) {
                const uint32_t at = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                const uint32_t sg = v[j] > 0.0f ? 1u : (v[j] < 0.0f ? 2u : 0u);
                s_cand[at] = ((unsigned long long)key[j] << 15) | ((unsigned long long)(8191u - idx) << 2) | sg;
            }
            base += (uint32_t)__popcll(m);
        }
    }
    __syncthreads();
    const uint32_t nc = s_ncand;       // <= kCand (n <= kCand: the whole frame)

    // ---- rank (ties: lower flat index first, Frame.m:176-178 on a stable sort): 2 threads per candidate ----
    {
        const uint32_t i = t & 127, part = t >> 7;
        if (i < nc) {
            const unsigned long long mine = s_cand[i];
            const uint32_t nblk = (nc + 7) >> 3, hblk = (nblk + 1) >> 1;
            const uint32_t b0 = part ? hblk : 0, b1 = part ? nblk : hblk;
            uint32_t r = 0;
            for (uint32_t blk = b0; blk < b1; ++blk) {
                const ulonglong2* q = reinterpret_cast<const ulonglong2*>(s_cand + 8 * blk);
                const ulonglong2 c0 = q[0], c1 = q[1], c2 = q[2], c3 = q[3];
                r += (c0.x > mine ? 1u : 0u) + (c0.y > mine ? 1u : 0u) + (c1.x > mine ? 1u : 0u) + (c1.y > mine ? 1u : 0u) +
                     (c2.x > mine ? 1u : 0u) + (c2.y > mine ? 1u : 0u) + (c3.x > mine ? 1u : 0u) + (c3.y > mine ? 1u : 0u);
            }
            if (r) atomicAdd(&s_rank[i], r);
        }
    }
    __syncthreads();
    if (t < (int)nc) {                 // sign pairs in rank order (Frame.m:182-190)
        const uint32_t rank = s_rank[t];
        if (rank < keep) {
            const uint32_t sg = (uint32_t)s_cand[t] & 3u;
            const uint32_t bpos = 2 * rank;
            if (sg == 1u) atomicOr(&s_bits[bpos >> 5], 1u << (bpos & 31));
            else if (sg == 2u && bpos + 1 < subfp_len) atomicOr(&s_bits[(bpos + 1) >> 5], 1u << ((bpos + 1) & 31));
        }
    }
    __syncthreads();
    if (t < (int)kPackedWords) packed[frame * kPackedWords + t] = s_bits[t];
}

}  // namespace

hipError_t launch_haar_select(const Plan& plan, float* d_frames, uint64_t n_frames, uint32_t* d_packed,
                              float* d_haar_out, hipStream_t stream) {
    if (n_frames == 0) return hipSuccess;
    if (n_frames > 0x7fffffffull) return hipErrorInvalidValue;
    size_t lds = (size_t)(5 * kRowsPerFrame * plan.bands / 2) * sizeof(float);            // frame, row-pass result, running sums
    if (lds < kHistBuckets * sizeof(uint32_t)) lds = kHistBuckets * sizeof(uint32_t);     // ... and the select's histogram afterwards
    const uint32_t per = (kRowsPerFrame * plan.bands + kThreads - 1) / kThreads;
    if (plan.keep > kCand || per > 32) return hipErrorInvalidValue;
    auto launch = [&](auto kern) -> hipError_t {
        static PerDevice attr;
        if (attr.changed(lds) && lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, dim3((uint32_t)n_frames), dim3(kThreads), lds, stream, d_frames, plan.bands,
                           std::sqrt((float)plan.bands), plan.keep, plan.subfp_len, d_packed, d_haar_out);
        return hipGetLastError();
    };
    if (per <= 8) return launch(haar_select_kernel<8>);
    if (per <= 16) return launch(haar_select_kernel<16>);
    return launch(haar_select_kernel<32>);
}

}  // namespace lbad
