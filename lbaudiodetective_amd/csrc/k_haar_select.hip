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
constexpr uint32_t kCandMax = 256;

__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t* s_red) {
    // wave reduce, then 4 partials through LDS
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    const int wave = threadIdx.x >> 6;
    __syncthreads();  // protect s_red from the previous use
    if ((threadIdx.x & 63) == 0) s_red[wave] = v;
    __syncthreads();
    return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// 1-D Haar of LBAudioDetectiveFrameDecomposeArray applied to `lines` independent lines of
// length `len`; element e of line l sits at src[l * lstride + e * estride].
// `root` = sqrtf(len) is computed on the host: the device's sqrt is the 1-ulp v_sqrt_f32 (sqrtf(14.0f)
// comes out one ulp low, for example), whatever -fhip-fp32-correctly-rounded-divide-sqrt promises.
__device__ void haar_lines(float* a, float* tmp, uint32_t lines, uint32_t len, uint32_t lstride, uint32_t estride,
                           float root) {
    const float root2 = __fsqrt_rn(2.0f);   // constant-folded by the compiler (correctly rounded)
    for (uint32_t p = threadIdx.x; p < lines * len; p += kThreads) {
        const uint32_t l = p / len, e = p % len;
        const uint32_t at = l * lstride + e * estride;
        a[at] = __fdiv_rn(a[at], root);
    }
    __syncthreads();
    uint32_t cnt = len;
    while (cnt > 1) {
        cnt >>= 1;
        for (uint32_t p = threadIdx.x; p < lines * cnt; p += kThreads) {
            const uint32_t l = p / cnt, i = p % cnt;
            const float ev = a[l * lstride + (2 * i) * estride];
            const float od = a[l * lstride + (2 * i + 1) * estride];
            tmp[l * lstride + i * estride] = __fdiv_rn(__fadd_rn(ev, od), root2);
            tmp[l * lstride + (cnt + i) * estride] = __fdiv_rn(__fsub_rn(ev, od), root2);
        }
        __syncthreads();
        for (uint32_t p = threadIdx.x; p < lines * 2 * cnt; p += kThreads) {
            const uint32_t l = p / (2 * cnt), i = p % (2 * cnt);
            const uint32_t at = l * lstride + i * estride;
            a[at] = tmp[at];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(kThreads) void haar_select_kernel(const float* __restrict__ frames, uint32_t bands,
                                                               float root_bands, uint32_t keep, uint32_t subfp_len,
                                                               uint32_t* __restrict__ packed,
                                                               float* __restrict__ haar_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const uint32_t n = kRowsPerFrame * bands;
    float* a = smem;
    float* tmp = smem + n;
    __shared__ uint32_t s_red[4];
    __shared__ uint32_t s_ckey[kCandMax];
    __shared__ uint32_t s_cidx[kCandMax];
    __shared__ uint32_t s_ncand;
    __shared__ uint32_t s_bits[kPackedWords];

    const uint64_t frame = blockIdx.x;
    const float* src = frames + frame * n;
    for (uint32_t i = threadIdx.x; i < n; i += kThreads) a[i] = src[i];
    if (threadIdx.x < kPackedWords) s_bits[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_ncand = 0;
    __syncthreads();

    haar_lines(a, tmp, kRowsPerFrame, bands, bands, 1, root_bands);                          // every row (Frame.m:114-116)
    haar_lines(a, tmp, bands, kRowsPerFrame, 1, bands, __fsqrt_rn((float)kRowsPerFrame));    // every column (:118-131); folded

    if (haar_out) {
        float* dst = haar_out + frame * n;
        for (uint32_t i = threadIdx.x; i < n; i += kThreads) dst[i] = a[i];
    }

    // ---- threshold search on key = |v| bits ------------------------------------------------
    uint32_t lo = 0, hi = 0x80000000u;  // count(key >= lo) >= keep > count(key >= hi)
    uint32_t cnt_lo = n;
    uint32_t idx_bound = n;             // among key == lo only flat indices < idx_bound are candidates
    if (n > kCandMax) {
        while (cnt_lo > kCandMax && hi - lo > 1) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            uint32_t c = 0;
            for (uint32_t i = threadIdx.x; i < n; i += kThreads)
                c += ((__float_as_uint(a[i]) & 0x7fffffffu) >= mid) ? 1u : 0u;
            c = block_sum(c, s_red);
            if (c >= keep) { lo = mid; cnt_lo = c; } else { hi = mid; }
        }
        if (cnt_lo > kCandMax) {
            // plateau: more than kCandMax coefficients share the threshold key `lo`.  Take every
            // key > lo, and of the equal ones the lowest flat indices until `keep` is reached.
            uint32_t g = 0;
            for (uint32_t i = threadIdx.x; i < n; i += kThreads)
                g += ((__float_as_uint(a[i]) & 0x7fffffffu) > lo) ? 1u : 0u;
            g = block_sum(g, s_red);
            uint32_t ilo = 0, ihi = n;  // smallest bound with g + ties(idx < bound) >= keep
            while (ilo < ihi) {
                const uint32_t im = ilo + ((ihi - ilo) >> 1);
                uint32_t c = 0;
                for (uint32_t i = threadIdx.x; i < n && i < im; i += kThreads)
                    c += ((__float_as_uint(a[i]) & 0x7fffffffu) == lo) ? 1u : 0u;
                c = block_sum(c, s_red);
                if (g + c >= keep) ihi = im; else ilo = im + 1;
            }
            idx_bound = ilo;
        }
    }

    // ---- gather candidates -----------------------------------------------------------------
    for (uint32_t i = threadIdx.x; i < n; i += kThreads) {
        const uint32_t key = __float_as_uint(a[i]) & 0x7fffffffu;
        if (key > lo || (key == lo && i < idx_bound)) {
            const uint32_t at = atomicAdd(&s_ncand, 1u);
            s_ckey[at] = key;
            s_cidx[at] = i;
        }
    }
    __syncthreads();
    const uint32_t nc = s_ncand;

    // ---- rank candidates among themselves, emit sign pairs in rank order (Frame.m:182-190) ----
    if (threadIdx.x < nc) {
        const uint32_t mk = s_ckey[threadIdx.x], mi = s_cidx[threadIdx.x];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < nc; ++j) {
            const uint32_t k = s_ckey[j], ii = s_cidx[j];
            rank += (k > mk || (k == mk && ii < mi)) ? 1u : 0u;
        }
        if (rank < keep) {
            const float v = a[mi];
            const uint32_t bpos = 2 * rank;
            if (v > 0.0f) {
                atomicOr(&s_bits[bpos >> 5], 1u << (bpos & 31));
            } else if (v < 0.0f && bpos + 1 < subfp_len) {
                atomicOr(&s_bits[(bpos + 1) >> 5], 1u << ((bpos + 1) & 31));
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < kPackedWords) packed[frame * kPackedWords + threadIdx.x] = s_bits[threadIdx.x];
}

}  // namespace

hipError_t launch_haar_select(const Plan& plan, float* d_frames, uint64_t n_frames, uint32_t* d_packed,
                              float* d_haar_out, hipStream_t stream) {
    if (n_frames == 0) return hipSuccess;
    if (n_frames > 0x7fffffffull) return hipErrorInvalidValue;
    const size_t lds = (size_t)2 * kRowsPerFrame * plan.bands * sizeof(float);
    hipLaunchKernelGGL(haar_select_kernel, dim3((uint32_t)n_frames), dim3(kThreads), lds, stream, d_frames,
                       plan.bands, std::sqrt((float)plan.bands), plan.keep, plan.subfp_len, d_packed, d_haar_out);
    return hipGetLastError();
}

}  // namespace lbad
