// k_fft_bands.hip -- unfused stage 1: analysis windows -> 32 band energies per window.
//
// Replaces the body of LBAudioDetectiveComputeFrequencies (LBAudioDetective.m:335-408) for every
// window of every clip: canonical radix-2 real FFT (same float32 operation order as
// oracle/lbad_oracle.c:rfft_exec), vDSP-style packing, band means with the reference's quirks
// (positive-only normalisation, divisor = edge difference).  One wave owns one window; the W/2
// complex points live in LDS in natural order (the decimation-in-time network is walked with
// bit-reversed index arithmetic instead of a bit-reversed load, so bin k ends at address
// brev(k)); three stages share one trip through LDS.  Generic over window size; the headline configuration uses k_rows_pruned.hip instead.
#include "internal.hpp"

namespace lbad {
namespace {

// Waves own disjoint LDS regions and never exchange data, so a workgroup barrier would only make
// them wait for each other.  LDS operations of one wave execute in order; the fences just stop the
// compiler from moving a lane's loads above other lanes' stores.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// complex point n lives at slot n + (n >> 5): one pad slot per 32 points keeps the strided gathers of
// the last passes (8 or 64 points between consecutive lanes) off a single bank group
__device__ __forceinline__ int zslot(int n) { return n + (n >> 5); }

// DIT stages S0 .. S0+NS-1 (NS <= 3) in one trip through LDS: a lane gathers the 2^NS points that
// these stages connect, runs the butterflies in registers and scatters the results back.  Points are
// stored in natural order; the stage-s partner of point n is n +- (N >> s) and its twiddle index is
// the bit-reversed value of n's top s-1 bits.  Every butterfly uses the general nested-fma form: for
// the twiddles 1 and -i this equals the oracle's multiplication-free form up to the sign of zeros.
template <int LOG2W, int S0, int NS>
__device__ __forceinline__ void dit_pass(float2* z, const float* __restrict__ twr, const float* __restrict__ twi,
                                         int lane) {
    constexpr int LOGN = LOG2W - 1;
    constexpr int N = 1 << LOGN;
    constexpr int G = 1 << NS;
    constexpr int low_bits = LOGN - S0 - NS + 1;        // bits of n below the NS varying ones
    constexpr int step = 1 << low_bits;                  // distance handled by the last stage of the pass
    for (int g = lane; g < N / G; g += 64) {
        const int n0 = ((g >> low_bits) << (low_bits + NS)) | (g & (step - 1));
        float2 x[G];
#pragma unroll
        for (int e = 0; e < G; ++e) x[e] = z[zslot(n0 + e * step)];
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            const int s = S0 + t;                        // global stage, partner distance N >> s
            const int half = G >> (t + 1);               // ... = `half` register slots
#pragma unroll
            for (int e = 0; e < G; ++e) {
                if (e & half) continue;
                const uint32_t n = (uint32_t)(n0 + e * step);
                const uint32_t j = s > 1 ? (__brev(n >> (LOGN - s + 1)) >> (32 - (s - 1))) : 0u;
                const uint32_t ti = j << (LOG2W - s);
                const float wr = twr[ti], wi = twi[ti];
                const float2 u = x[e], v = x[e + half];
                x[e].x = __fmaf_rn(wr, v.x, __fmaf_rn(-wi, v.y, u.x));
                x[e].y = __fmaf_rn(wr, v.y, __fmaf_rn(wi, v.x, u.y));
                x[e + half].x = __fmaf_rn(-wr, v.x, __fmaf_rn(wi, v.y, u.x));
                x[e + half].y = __fmaf_rn(-wr, v.y, __fmaf_rn(-wi, v.x, u.y));
            }
        }
#pragma unroll
        for (int e = 0; e < G; ++e) z[zslot(n0 + e * step)] = x[e];
    }
    wave_sync();
}

template <int LOG2W, int S0>
__device__ __forceinline__ void dit_all(float2* z, const float* __restrict__ twr, const float* __restrict__ twi,
                                        int lane) {
    constexpr int LOGN = LOG2W - 1;
    if constexpr (S0 <= LOGN) {
        constexpr int left = LOGN - S0 + 1;
        constexpr int NS = left >= 3 ? 3 : left;
        dit_pass<LOG2W, S0, NS>(z, twr, twi, lane);
        dit_all<LOG2W, S0 + NS>(z, twr, twi, lane);
    }
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
    // per wave: N (+ skew) complex points, then N floats of power terms
    constexpr int kZf = 2 * (N + (N >> 5));
    float* zf = smem + (size_t)wave * (kZf + N);
    float2* z = reinterpret_cast<float2*>(zf);
    float* vbuf = zf + kZf;

    const uint64_t win = (uint64_t)blockIdx.x * WPB + wave;
    const uint64_t clip = win / windows_per_clip;
    const uint32_t wi = (uint32_t)(win % windows_per_clip);
    const uint64_t first = clip * samples_per_clip + (uint64_t)wi * stride;
    // sample formats: 0 float32, 1 int16 / 32768, 2 int32 / 2^31 (what LBAudioDetectiveConvertToFormat,
    // LBAudioDetective.m:413-437, asks AudioConverter to do for integer PCM)
    if (fmt == 0) {
        const float* src = static_cast<const float*>(pcm_raw) + first;
        for (int i = lane; i < W; i += 64) zf[2 * zslot(i >> 1) + (i & 1)] = src[i];
    } else if (fmt == 1) {
        const int16_t* src = static_cast<const int16_t*>(pcm_raw) + first;
        for (int i = lane; i < W; i += 64) zf[2 * zslot(i >> 1) + (i & 1)] = (float)src[i] * (1.0f / 32768.0f);
    } else {
        const int32_t* src = static_cast<const int32_t*>(pcm_raw) + first;
        for (int i = lane; i < W; i += 64) zf[2 * zslot(i >> 1) + (i & 1)] = (float)src[i] * (1.0f / 2147483648.0f);
    }
    wave_sync();

    const float* twr = tw;   // gathered by bit-reversed index; L1/L2 resident
    const float* twi = tw + N;
    dit_all<LOG2W, 1>(z, twr, twi, lane);

    // split pass for the bins the bands read, then the reference's per-bin power term
    const float inv_norm = 1.0f / (float)(W / 4);  // (Float32)(width/2), width = W/2; exact power of two
    for (uint32_t k = kmin + lane; k < kmax; k += 64) {
        float re, im;
        if (k == 0) {
            const float2 z0 = z[0];   // zslot(0) == 0
            const float sm = z0.x + z0.y, df = z0.x - z0.y;
            re = sm + sm;
            im = df + df;
        } else {
            const float2 a = z[zslot((int)(__brev(k) >> (32 - LOGN)))];
            const float2 b = z[zslot((int)(__brev((uint32_t)N - k) >> (32 - LOGN)))];
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
    wave_sync();

    for (uint32_t b = lane; b < nbands; b += 64) {
        const uint32_t lo = band_tbl[b], hi = band_tbl[nbands + b];
        const float div = __uint_as_float(band_tbl[2 * nbands + b]);
        // the sum must run in bin order (float32 addition is not associative); the loads are issued in
        // batches of 8 so that their LDS latency overlaps instead of serialising with the adds
        float p = 0.0f;
        for (uint32_t k0 = lo; k0 < hi; k0 += 8) {
            float x[8];
#pragma unroll
            for (uint32_t q = 0; q < 8; ++q) x[q] = (k0 + q < hi) ? vbuf[k0 + q - kmin] : 0.0f;
#pragma unroll
            for (uint32_t q = 0; q < 8; ++q)
                if (k0 + q < hi && x[q] == x[q] && fabsf(x[q]) != INFINITY) p = __fadd_rn(p, x[q]);
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
    const size_t lds = (size_t)WPB * (2 * (W / 2 + W / 64) + W / 2) * sizeof(float);
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
