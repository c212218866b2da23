// k_fft_bands.hip -- unfused stage 1: analysis windows -> 32 band energies per window.
//
// Replaces the body of LBAudioDetectiveComputeFrequencies (LBAudioDetective.m:335-408) for every
// window of every clip: canonical radix-2 real FFT (same float32 operation order as
// oracle/lbad_oracle.c:rfft_exec), vDSP-style packing, band means with the reference's quirks
// (positive-only normalisation, divisor = edge difference).  One wave owns one window; the W/2
// complex points live in LDS in natural order (the decimation-in-time network is walked with
// bit-reversed index arithmetic instead of a bit-reversed load, so bin k ends at address
// brev(k)).  Generic over window size; the headline configuration uses k_rows_pruned.hip instead.
#include "internal.hpp"

namespace lbad {
namespace {

__device__ __forceinline__ float2 bfly_mul_add(float wr, float wi, float2 u, float2 v, float2& lo) {
    // out0 = u + w v, out1 = u - w v; nested fma, identical to the oracle's bfly_general
    float2 hi;
    hi.x = __fmaf_rn(wr, v.x, __fmaf_rn(-wi, v.y, u.x));
    hi.y = __fmaf_rn(wr, v.y, __fmaf_rn(wi, v.x, u.y));
    lo.x = __fmaf_rn(-wr, v.x, __fmaf_rn(wi, v.y, u.x));
    lo.y = __fmaf_rn(-wr, v.y, __fmaf_rn(-wi, v.x, u.y));
    return hi;
}

template <int LOG2W, int WPB>
__global__ __launch_bounds__(WPB * 64) void fft_bands_kernel(
    const void* __restrict__ pcm_raw, uint32_t fmt, uint64_t samples_per_clip, uint32_t stride,
    uint32_t windows_per_clip, const float* __restrict__ tw, const uint32_t* __restrict__ band_tbl, uint32_t nbands, uint32_t kmin,
    uint32_t kmax, float* __restrict__ frames) {
    constexpr int W = 1 << LOG2W;
    constexpr int N = W / 2;
    constexpr int LOGN = LOG2W - 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    float* zf = smem + (size_t)wave * (W + N);
    float2* z = reinterpret_cast<float2*>(zf);
    float* vbuf = zf + W;

    const uint64_t win = (uint64_t)blockIdx.x * WPB + wave;
    const uint64_t clip = win / windows_per_clip;
    const uint32_t wi = (uint32_t)(win % windows_per_clip);
    const uint64_t first = clip * samples_per_clip + (uint64_t)wi * stride;
    // sample formats: 0 float32, 1 int16 / 32768, 2 int32 / 2^31 (what LBAudioDetectiveConvertToFormat,
    // LBAudioDetective.m:413-437, asks AudioConverter to do for integer PCM)
    if (fmt == 0) {
        const float* src = static_cast<const float*>(pcm_raw) + first;
        for (int i = lane; i < W; i += 64) zf[i] = src[i];
    } else if (fmt == 1) {
        const int16_t* src = static_cast<const int16_t*>(pcm_raw) + first;
        for (int i = lane; i < W; i += 64) zf[i] = (float)src[i] * (1.0f / 32768.0f);
    } else {
        const int32_t* src = static_cast<const int32_t*>(pcm_raw) + first;
        for (int i = lane; i < W; i += 64) zf[i] = (float)src[i] * (1.0f / 2147483648.0f);
    }
    __syncthreads();

    const float* twr = tw;
    const float* twi = tw + N;

#pragma unroll
    for (int s = 1; s <= LOGN; ++s) {
        const int logD = LOGN - s;
        const int D = 1 << logD;
        for (int b = lane; b < N / 2; b += 64) {
            const int n0 = ((b >> logD) << (logD + 1)) | (b & (D - 1));
            const int n1 = n0 + D;
            const uint32_t j = (s > 1) ? (__brev((uint32_t)(n0 >> (logD + 1))) >> (32 - (s - 1))) : 0u;
            const float2 u = z[n0], v = z[n1];
            float2 o0, o1;
            if (j == 0) {
                o0 = make_float2(u.x + v.x, u.y + v.y);
                o1 = make_float2(u.x - v.x, u.y - v.y);
            } else if ((j << 2) == (1u << s)) {  // w = -i
                o0 = make_float2(u.x + v.y, u.y - v.x);
                o1 = make_float2(u.x - v.y, u.y + v.x);
            } else {
                const uint32_t t = j << (LOG2W - s);
                o0 = bfly_mul_add(twr[t], twi[t], u, v, o1);
            }
            z[n0] = o0;
            z[n1] = o1;
        }
        __syncthreads();
    }

    // split pass for the bins the bands read, then the reference's per-bin power term
    const float inv_norm = 1.0f / (float)(W / 4);  // (Float32)(width/2), width = W/2; exact power of two
    for (uint32_t k = kmin + lane; k < kmax; k += 64) {
        float re, im;
        if (k == 0) {
            const float2 z0 = z[0];
            const float sm = z0.x + z0.y, df = z0.x - z0.y;
            re = sm + sm;
            im = df + df;
        } else {
            const float2 a = z[__brev(k) >> (32 - LOGN)];
            const float2 b = z[__brev((uint32_t)N - k) >> (32 - LOGN)];
            const float sr = a.x + b.x, si = a.y - b.y;
            const float dr = a.x - b.x, di = a.y + b.y;
            const float wr = twr[k], wi2 = twi[k];
            re = __fmaf_rn(wr, di, __fmaf_rn(wi2, dr, sr));
            im = __fmaf_rn(-wr, dr, __fmaf_rn(wi2, di, si));
        }
        if (re > 0.0f) re = __fmul_rn(re, inv_norm);
        if (im > 0.0f) im = __fmul_rn(im, inv_norm);
        vbuf[k - kmin] = __fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im));
    }
    __syncthreads();

    for (uint32_t b = lane; b < nbands; b += 64) {
        const uint32_t lo = band_tbl[b], hi = band_tbl[nbands + b];
        const float div = __uint_as_float(band_tbl[2 * nbands + b]);
        float p = 0.0f;
        for (uint32_t k = lo; k < hi; ++k) {
            const float x = vbuf[k - kmin];
            if (x == x && fabsf(x) != INFINITY) p = __fadd_rn(p, x);
        }
        frames[win * nbands + b] = __fdiv_rn(p, div);
    }
}

template <int LOG2W>
hipError_t launch_one(const Plan& plan, const void* d_pcm, uint32_t fmt, uint64_t n_clips, uint64_t samples_per_clip,
                      uint32_t frames_per_clip, float* d_frames, hipStream_t stream) {
    constexpr int W = 1 << LOG2W;
    constexpr int WPB = (W <= 4096) ? 4 : 2;
    const uint32_t windows_per_clip = frames_per_clip * kRowsPerFrame;
    const uint64_t n_windows = n_clips * windows_per_clip;
    if (n_windows == 0) return hipSuccess;
    const uint64_t blocks = n_windows / WPB;  // windows come in multiples of 128
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    const size_t lds = (size_t)WPB * (W + W / 2) * sizeof(float);
    auto kern = fft_bands_kernel<LOG2W, WPB>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((uint32_t)blocks), dim3(WPB * 64), lds, stream, d_pcm, fmt, samples_per_clip,
                       plan.stride, windows_per_clip, plan.d_tw, plan.d_bands, plan.bands, plan.table.kmin,
                       plan.table.kmax, d_frames);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_fft_bands(const Plan& plan, const void* d_pcm, uint32_t fmt, uint64_t n_clips,
                            uint64_t samples_per_clip, uint32_t frames_per_clip, float* d_frames, hipStream_t stream) {
    switch (plan.log2w) {
        case 4: return launch_one<4>(plan, d_pcm, fmt, n_clips, samples_per_clip, frames_per_clip, d_frames, stream);
        case 5: return launch_one<5>(plan, d_pcm, fmt, n_clips, samples_per_clip, frames_per_clip, d_frames, stream);
        case 6: return launch_one<6>(plan, d_pcm, fmt, n_clips, samples_per_clip, frames_per_clip, d_frames, stream);
        case 7: return launch_one<7>(plan, d_pcm, fmt, n_clips, samples_per_clip, frames_per_clip, d_frames, stream);
        case 8: return launch_one<8>(plan, d_pcm, fmt, n_clips, samples_per_clip, frames_per_clip, d_frames, stream);
        case 9: return launch_one<9>(plan, d_pcm, fmt, n_clips, samples_per_clip, frames_per_clip, d_frames, stream);
        case 10: return launch_one<10>(plan, d_pcm, fmt, n_clips, samples_per_clip, frames_per_clip, d_frames, stream);
        case 11: return launch_one<11>(plan, d_pcm, fmt, n_clips, samples_per_clip, frames_per_clip, d_frames, stream);
        case 12: return launch_one<12>(plan, d_pcm, fmt, n_clips, samples_per_clip, frames_per_clip, d_frames, stream);
        case 13: return launch_one<13>(plan, d_pcm, fmt, n_clips, samples_per_clip, frames_per_clip, d_frames, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace lbad
