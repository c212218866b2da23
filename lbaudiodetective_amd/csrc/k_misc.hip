// k_misc.hip -- small kernels behind the Frame API and the synthetic-input generators.
#include "internal.hpp"

#include <cmath>

namespace lbad {
namespace {

constexpr int kThreads = 256;

// ---- generic 2-D Haar (LBAudioDetectiveFrame.m:113-153) for arbitrary rows x cols -------------
// One thread walks one line serially, exactly like LBAudioDetectiveFrameDecomposeArray; used
// only by the Frame API (the known-answer test matrix is 3 x 4), never by the batch path.
// `root` = sqrtf(len) comes from the host: the device's sqrt is the 1-ulp v_sqrt_f32 (sqrtf(14.0f) is one
// ulp low, for example), whatever -fhip-fp32-correctly-rounded-divide-sqrt promises.
__global__ __launch_bounds__(kThreads) void haar_lines_kernel(float* m, float* tmp, uint32_t lines, uint32_t len,
                                                              uint32_t lstride, uint32_t estride, float root) {
    const uint32_t l = blockIdx.x * kThreads + threadIdx.x;
    if (l >= lines) return;
    float* a = m + (size_t)l * lstride;
    float* t = tmp + (size_t)l * lstride;
    const float root2 = __fsqrt_rn(2.0f);   // constant-folded by the compiler (correctly rounded)
    for (uint32_t i = 0; i < len; ++i) a[(size_t)i * estride] = __fdiv_rn(a[(size_t)i * estride], root);
    uint32_t cnt = len;
    while (cnt > 1) {
        cnt >>= 1;
        for (uint32_t i = 0; i < cnt; ++i) {
            const float ev = a[(size_t)(2 * i) * estride], od = a[(size_t)(2 * i + 1) * estride];
            t[(size_t)i * estride] = __fdiv_rn(__fadd_rn(ev, od), root2);
            t[(size_t)(cnt + i) * estride] = __fdiv_rn(__fsub_rn(ev, od), root2);
        }
        for (uint32_t i = 0; i < 2 * cnt; ++i) a[(size_t)i * estride] = t[(size_t)i * estride];
    }
}

// ---- generic ranked sign extraction (LBAudioDetectiveFrame.m:165-191) --------------------------
// rank(i) = #{ j : |v_j| > |v_i|  or  (|v_j| == |v_i| and j < i) }; ranks below n_wavelets emit
// their sign pair at out[2 rank], out[2 rank + 1].
__global__ __launch_bounds__(kThreads) void extract_kernel(const float* __restrict__ m, uint32_t n,
                                                           uint32_t n_wavelets, uint8_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    __shared__ uint32_t s_keys[kThreads];
    const uint32_t mine = i < n ? (__float_as_uint(m[i]) & 0x7fffffffu) : 0u;
    uint32_t rank = 0;
    for (uint32_t base = 0; base < n; base += kThreads) {
        const uint32_t j = base + threadIdx.x;
        __syncthreads();
        s_keys[threadIdx.x] = j < n ? (__float_as_uint(m[j]) & 0x7fffffffu) : 0u;
        __syncthreads();
        const uint32_t lim = (n - base) < (uint32_t)kThreads ? (n - base) : (uint32_t)kThreads;
        for (uint32_t t = 0; t < lim; ++t) {
            const uint32_t k = s_keys[t];
            rank += (k > mine || (k == mine && base + t < i)) ? 1u : 0u;
        }
    }
    if (i < n && rank < n_wavelets) {
        const float v = m[i];
        if (v > 0.0f) out[2 * rank] = 1;
        else if (v < 0.0f) out[2 * rank + 1] = 1;
    }
}

// ---- synthetic inputs; integer arithmetic identical to oracle/lbad_oracle.c --------------------
__device__ __host__ inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

__constant__ int16_t c_sine[1024];

__device__ __forceinline__ int32_t synth_channel(uint32_t key, const uint32_t* step, const uint32_t* phase0,
                                                 const int32_t* amp, uint32_t n) {
    int32_t acc = (int32_t)(mix32(key ^ (n * 0x9E3779B1u)) >> 18) - 8192;
#pragma unroll
    for (uint32_t j = 0; j < 3; ++j) {
        const uint32_t ph = phase0[j] + n * step[j];
        acc += (amp[j] * (int32_t)c_sine[ph >> 22]) >> 15;
    }
    acc = acc > 32767 ? 32767 : acc;
    acc = acc < -32768 ? -32768 : acc;
    return acc;
}

__device__ __forceinline__ void channel_params(uint32_t key, uint32_t rate_hz, uint32_t* step, uint32_t* phase0,
                                               int32_t* amp) {
#pragma unroll
    for (uint32_t j = 0; j < 3; ++j) {
        const uint32_t f_mhz = 60000u + mix32(key + 11u * j + 1u) % 840001u;
        step[j] = (uint32_t)((((uint64_t)f_mhz) << 32) / ((uint64_t)rate_hz * 1000u));
        phase0[j] = mix32(key + 11u * j + 2u);
        amp[j] = 1638 + (int32_t)(mix32(key + 11u * j + 3u) % 8193u);
    }
}

__global__ __launch_bounds__(kThreads) void synth_clips_kernel(uint32_t seed, uint64_t first, uint32_t rate_hz,
                                                               uint32_t n_samples, uint32_t stereo,
                                                               float* __restrict__ out) {
    const uint64_t clip = first + blockIdx.y;
    const uint32_t key = mix32(seed ^ mix32((uint32_t)clip) ^ (uint32_t)(clip >> 32) * 0x632BE5ABu);
    uint32_t step[3], phase0[3], step_r[3], phase_r[3];
    int32_t amp[3], amp_r[3];
    channel_params(key, rate_hz, step, phase0, amp);
    if (stereo) channel_params(key ^ 0x5bd1e995u, rate_hz, step_r, phase_r, amp_r);
    float* dst = out + (uint64_t)blockIdx.y * n_samples;
    for (uint32_t n = blockIdx.x * kThreads + threadIdx.x; n < n_samples; n += gridDim.x * kThreads) {
        if (stereo) {
            const int32_t l = synth_channel(key, step, phase0, amp, n);
            const int32_t r = synth_channel(key ^ 0x5bd1e995u, step_r, phase_r, amp_r, n);
            dst[n] = (float)(l + r) / 65536.0f;
        } else {
            dst[n] = (float)synth_channel(key, step, phase0, amp, n) / 32768.0f;
        }
    }
}

__global__ __launch_bounds__(kThreads) void synth_corpus_kernel(uint32_t seed, uint64_t first, uint64_t n_entries,
                                                                uint32_t n_sub, uint32_t subfp_len,
                                                                uint32_t* __restrict__ out) {
    // one thread per (entry, sub-fingerprint)
    const uint64_t t = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    if (t >= n_entries * n_sub) return;
    const uint64_t entry = first + t / n_sub;
    const uint32_t s = (uint32_t)(t % n_sub);
    const uint32_t key = mix32(seed ^ mix32((uint32_t)entry) ^ (uint32_t)(entry >> 32) * 0x632BE5ABu);
    const uint32_t pairs = (subfp_len + 1) / 2;
    uint32_t w[kPackedWords] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 4
    for (uint32_t p = 0; p < pairs; ++p) {
        const uint32_t r = mix32(key + (s * 1024u + p) * 0x9E3779B1u);
        uint32_t pos = 0, neg = 0;
        if (r % 100u != 0u) {
            if ((r >> 8) & 1u) pos = 1; else neg = 1;
        }
        const uint32_t b = 2 * p;
        if (b + 1 >= subfp_len) neg = 0;
        const uint32_t two = pos | (neg << 1);
        // dynamic word index on a small local array: keep it branch-free
#pragma unroll
        for (uint32_t k = 0; k < kPackedWords; ++k)
            if (k == (b >> 5)) w[k] |= two << (b & 31);
    }
    uint4* dst = reinterpret_cast<uint4*>(out + t * kPackedWords);
    dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
    dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

void sine_table(int16_t* t) {
    for (int i = 0; i <= 256; ++i) {
        const double s = sin((2.0 * M_PI * i) / 1024.0) * 32767.0;
        const int16_t v = (int16_t)floor(s + 0.5);
        t[i] = v;
        if (i > 0 && i < 256) {
            t[512 - i] = v;
            t[512 + i] = (int16_t)-v;
            t[1024 - i] = (int16_t)-v;
        }
    }
    t[512] = 0;
    t[768] = (int16_t)-t[256];
}

}  // namespace

hipError_t launch_haar2d_generic(float* d_m, float* d_tmp, uint32_t rows, uint32_t cols, hipStream_t stream) {
    if (rows == 0 || cols == 0) return hipSuccess;
    hipLaunchKernelGGL(haar_lines_kernel, dim3((rows + kThreads - 1) / kThreads), dim3(kThreads), 0, stream, d_m,
                       d_tmp, rows, cols, cols, 1u, std::sqrt((float)cols));
    hipLaunchKernelGGL(haar_lines_kernel, dim3((cols + kThreads - 1) / kThreads), dim3(kThreads), 0, stream, d_m,
                       d_tmp, cols, rows, 1u, cols, std::sqrt((float)rows));
    return hipGetLastError();
}

hipError_t launch_extract_generic(const float* d_m, uint32_t n, uint32_t n_wavelets, uint8_t* d_out,
                                  hipStream_t stream) {
    hipError_t e = hipMemsetAsync(d_out, 0, (size_t)2 * n_wavelets, stream);
    if (e != hipSuccess || n == 0) return e;
    hipLaunchKernelGGL(extract_kernel, dim3((n + kThreads - 1) / kThreads), dim3(kThreads), 0, stream, d_m, n,
                       n_wavelets, d_out);
    return hipGetLastError();
}

static hipError_t ensure_sine() {
    static PerDevice done;            // __constant__ memory is per device
    if (!done.changed(1)) return hipSuccess;
    int16_t t[1024];
    sine_table(t);
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_sine), t, sizeof(t));
    if (e != hipSuccess) done.changed(0);
    return e;
}

hipError_t launch_synth_clips(uint32_t seed, uint64_t first, uint64_t n_clips, uint32_t rate_hz, uint32_t n_samples,
                              uint32_t stereo, float* d_out, hipStream_t stream) {
    if (n_clips == 0 || n_samples == 0) return hipSuccess;
    hipError_t e = ensure_sine();
    if (e != hipSuccess) return e;
    uint32_t bx = (n_samples + kThreads - 1) / kThreads;
    if (bx > 64) bx = 64;
    // grid.y is limited to 65535: chunk the clips
    for (uint64_t done = 0; done < n_clips;) {
        const uint64_t chunk = (n_clips - done) < 65535ull ? (n_clips - done) : 65535ull;
        hipLaunchKernelGGL(synth_clips_kernel, dim3(bx, (uint32_t)chunk), dim3(kThreads), 0, stream, seed,
                           first + done, rate_hz, n_samples, stereo, d_out + done * n_samples);
        done += chunk;
    }
    return hipGetLastError();
}

// Shader clock under whatever load the device carries right now: one wave spins for about `usec` microseconds and
// reports how far s_memtime (shader-clock ticks) and s_memrealtime (a constant 100 MHz) advanced.  Meant to be
// launched on a side stream next to a running kernel (it needs one wave slot, no LDS).
__global__ __launch_bounds__(64) void clock_probe_kernel(uint32_t usec, unsigned long long* __restrict__ out) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < (unsigned long long)usec * 100ull) {
        __builtin_amdgcn_s_sleep(8);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[0] = t1 - t0;
        out[1] = r1 - r0;
    }
}

hipError_t launch_clock_probe(uint32_t usec, unsigned long long* d_out, hipStream_t stream) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, stream, usec, d_out);
    return hipGetLastError();
}

hipError_t launch_synth_corpus(uint32_t seed, uint64_t first, uint64_t n_entries, uint32_t n_sub, uint32_t subfp_len,
                               uint32_t* d_out, hipStream_t stream) {
    const uint64_t total = n_entries * n_sub;
    if (total == 0) return hipSuccess;
    const uint64_t blocks = (total + kThreads - 1) / kThreads;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(synth_corpus_kernel, dim3((uint32_t)blocks), dim3(kThreads), 0, stream, seed, first,
                       n_entries, n_sub, subfp_len, d_out);
    return hipGetLastError();
}

}  // namespace lbad
