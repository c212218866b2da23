// k_fft_bands.hip -- unfused stage 1: analysis windows -> 32 band energies per window.
//
// Replaces the body of LBAudioDetectiveComputeFrequencies (LBAudioDetective.m:335-408) for every
// window of every clip: canonical radix-2 real FFT (same float32 operation order as
// oracle/lbad_oracle.c:rfft_exec), vDSP-style packing, band means with the reference's quirks
// (positive-only normalisation, divisor = edge difference).  One wave owns one window; the W/2
// complex points live in LDS in natural order (the decimation-in-time network is walked with
// bit-reversed index arithmetic instead of a bit-reversed load, so bin k ends at address
// brev(k)); three stages share one trip through LDS and persistent workgroups keep every lane's
// twiddles in an LDS cache laid out in consumption order.  Generic over window size; the headline configuration uses k_rows_pruned.hip instead.
#include "internal.hpp"

#include <cstdlib>

namespace lbad {
namespace {

// complex value = one even-aligned VGPR pair; butterflies on the pair compile to v_pk_fma_f32 with
// op_sel swizzles (each half is the same correctly rounded fma the oracle performs)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 mk2(float x, float y) {
    f32x2 r;
    r.x = x;
    r.y = y;
    return r;
}

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

// Pass structure of the N = W/2 point DIT network: stages are taken three at a time (the last pass
// takes what is left).  Helpers below are constexpr so that LDS offsets are compile-time constants.
template <int LOG2W> struct Passes {
    static constexpr int LOGN = LOG2W - 1;
    static constexpr int N = 1 << LOGN;
    // Pass 0 has wave-uniform twiddles and may therefore take more stages per LDS trip than the others:
    // five at W = 4096 (32 points per lane, passes 5 + 3 + 3 instead of 3 + 3 + 3 + 2), three elsewhere.
    static constexpr int kFirst = LOG2W == 12 ? 5 : (LOGN >= 3 ? 3 : LOGN);
    static constexpr int count = 1 + (LOGN - kFirst + 2) / 3;
    static constexpr int first_stage(int p) { return p == 0 ? 1 : 1 + kFirst + 3 * (p - 1); }
    static constexpr int stages(int p) {
        return p == 0 ? kFirst : ((LOGN - kFirst - 3 * (p - 1)) >= 3 ? 3 : (LOGN - kFirst - 3 * (p - 1)));
    }
    static constexpr int groups_per_lane(int p) { return ((N >> stages(p)) + 63) / 64; }
    static constexpr int twiddles(int p) { return (1 << stages(p)) - 1; }
    // float2 slots of the per-lane twiddle cache before pass p (pass 0 has wave-uniform twiddles: the
    // eighth roots of unity, kept in scalar registers instead)
    static constexpr int cache_offset(int p) {
        int o = 0;
        for (int i = 1; i < p; ++i) o += groups_per_lane(i) * twiddles(i) * 64;
        return o;
    }
    static constexpr int cache_slots = cache_offset(count);
};

// Per-lane twiddle cache.  The butterfly (pass p, group slot q, twiddle t) of lane l always needs the
// same table entry, whatever the window, so a persistent workgroup gathers them once from the master
// table into LDS in exactly the order the lanes consume them: cache[p][q][t][l] -- conflict-free
// 64-bit reads instead of ~130 scattered global loads per window.
template <int LOG2W, int P>
__device__ __forceinline__ void build_cache(float2* cache, const float* __restrict__ tw, int tid, int nthreads) {
    using Ps = Passes<LOG2W>;
    if constexpr (P == 0) {
        build_cache<LOG2W, 1>(cache, tw, tid, nthreads);
    } else if constexpr (P < Ps::count) {
        constexpr int LOGN = Ps::LOGN, N = Ps::N;
        constexpr int S0 = Ps::first_stage(P), NS = Ps::stages(P), G = 1 << NS;
        constexpr int low_bits = LOGN - S0 - NS + 1, step = 1 << low_bits;
        constexpr int GP = Ps::groups_per_lane(P), T = Ps::twiddles(P);
        for (int i = tid; i < GP * T * 64; i += nthreads) {
            const int lane = i & 63, t_id = (i >> 6) % T, q = (i >> 6) / T;
            const int g = lane + 64 * q;
            float2 w = make_float2(1.0f, 0.0f);
            if (g < N / G) {
                // t_id = (2^t - 1) + h: stage t of the pass, h = the group's register slot index / (2 * half)
                int t = 0;
                while ((2 << t) - 1 <= t_id) ++t;
                const int h = t_id - ((1 << t) - 1);
                const int s = S0 + t;
                const int e = h << (NS - t);
                const int n0 = ((g >> low_bits) << (low_bits + NS)) | (g & (step - 1));
                const uint32_t n = (uint32_t)(n0 + e * step);
                const uint32_t j = s > 1 ? (__brev(n >> (LOGN - s + 1)) >> (32 - (s - 1))) : 0u;
                const uint32_t ti = j << (LOG2W - s);
                w = make_float2(tw[ti], tw[N + ti]);
            }
            cache[Ps::cache_offset(P) + i] = w;
        }
        build_cache<LOG2W, P + 1>(cache, tw, tid, nthreads);
    }
}

// DIT stages of pass P in one trip through LDS: a lane gathers the 2^NS points these stages connect,
// runs the butterflies in registers and scatters the results back.  Points are stored in natural
// order; the stage-s partner of point n is n +- (N >> s) and its twiddle index is the bit-reversed
// value of n's top s-1 bits.  Every butterfly uses the general nested-fma form: for the twiddles 1 and
// -i this equals the oracle's multiplication-free form up to the sign of zeros.
// Pass 0 (stages 1..kFirst): the twiddle of a butterfly depends only on its register slot --
// W_(2^(t+1))^j with j the bit-reversed block index -- so the few distinct values are read once from the
// master table into scalar registers, stages 1 and 2 (and every j = 0 or quarter-turn butterfly) are the
// oracle's multiplication-free forms, and no per-lane twiddle is fetched at all.
__device__ __forceinline__ constexpr int brev_bits(int v, int bits) {
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1) << (bits - 1 - i);
    return r;
}

template <int LOG2W>
__device__ __forceinline__ void dit_pass0(float2* z, const float* __restrict__ tw, int lane) {
    using Ps = Passes<LOG2W>;
    constexpr int N = Ps::N;
    constexpr int NS = Ps::stages(0), G = 1 << NS;
    constexpr int step = N >> NS;
    // distinct non-trivial twiddles: W_(2^NS)^j for j = 1 .. 2^(NS-1) - 1 (index j << (LOG2W - NS)); the
    // stage-t twiddle W_(2^(t+1))^j is entry j << (NS - 1 - t) of the same family
    constexpr int NT = NS >= 1 ? (1 << (NS - 1)) : 1;
    float wre[NT], wim[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        wre[j] = tw[j << (LOG2W - NS)];
        wim[j] = tw[N + (j << (LOG2W - NS))];
    }
#pragma unroll 2
    for (int q = 0; q < Ps::groups_per_lane(0); ++q) {
        const int g = lane + 64 * q;
        if (g < step) {
            f32x2 x[G];
#pragma unroll
            for (int e = 0; e < G; ++e) {
                const float2 t = z[zslot(g + e * step)];
                x[e] = mk2(t.x, t.y);
            }
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                const int half = G >> (t + 1);
#pragma unroll
                for (int e = 0; e < G; ++e) {
                    if (e & half) continue;
                    const int h = e >> (NS - t);                       // butterfly block of this stage
                    const int j = brev_bits(h, t);                     // twiddle W_(2^(t+1))^j
                    const int fam = j << (NS - 1 - t);                 // its index in the W_(2^NS) family
                    const f32x2 u = x[e], v = x[e + half], vs = v.yx;
                    if (fam == 0) {                                     // w = 1
                        x[e] = u + v;
                        x[e + half] = u - v;
                    } else if (2 * fam == NT) {                         // w = -i: w v = (v.y, -v.x)
                        x[e] = fma2(mk2(1.0f, -1.0f), vs, u);
                        x[e + half] = fma2(mk2(-1.0f, 1.0f), vs, u);
                    } else {
                        const float wr = wre[fam], wi = wim[fam];
                        x[e] = fma2(mk2(wr, wr), v, fma2(mk2(-wi, wi), vs, u));
                        x[e + half] = fma2(mk2(-wr, -wr), v, fma2(mk2(wi, -wi), vs, u));
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < G; ++e) z[zslot(g + e * step)] = make_float2(x[e].x, x[e].y);
        }
    }
    wave_sync();
}

template <int LOG2W, int P, bool CACHED>
__device__ __forceinline__ void dit_pass(float2* z, const float2* cache, const float* __restrict__ tw, int lane) {
    using Ps = Passes<LOG2W>;
    constexpr int LOGN = Ps::LOGN, N = Ps::N;
    constexpr int S0 = Ps::first_stage(P), NS = Ps::stages(P), G = 1 << NS;
    constexpr int low_bits = LOGN - S0 - NS + 1;        // bits of n below the NS varying ones
    constexpr int step = 1 << low_bits;                  // distance handled by the last stage of the pass
    constexpr int T = Ps::twiddles(P);
    // two groups of a lane are in flight together (their LDS round trips overlap); unrolling them all
    // only multiplies the live registers (8 points + 7 twiddles each)
#pragma unroll 2
    for (int q = 0; q < Ps::groups_per_lane(P); ++q) {
        const int g = lane + 64 * q;
        if (g < N / G) {
            const int n0 = ((g >> low_bits) << (low_bits + NS)) | (g & (step - 1));
            float2 w[T];
            if constexpr (CACHED) {
#pragma unroll
                for (int t_id = 0; t_id < T; ++t_id) w[t_id] = cache[Ps::cache_offset(P) + (q * T + t_id) * 64 + lane];
            } else {
#pragma unroll
                for (int t = 0; t < NS; ++t)
#pragma unroll
                    for (int h = 0; h < (1 << t); ++h) {
                        const int s = S0 + t;
                        const uint32_t n = (uint32_t)(n0 + (h << (NS - t)) * step);
                        const uint32_t j = s > 1 ? (__brev(n >> (LOGN - s + 1)) >> (32 - (s - 1))) : 0u;
                        const uint32_t ti = j << (LOG2W - s);
                        w[(1 << t) - 1 + h] = make_float2(tw[ti], tw[N + ti]);
                    }
            }
            f32x2 x[G];
#pragma unroll
            for (int e = 0; e < G; ++e) {
                const float2 t = z[zslot(n0 + e * step)];
                x[e] = mk2(t.x, t.y);
            }
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                const int half = G >> (t + 1);               // partner distance in register slots
#pragma unroll
                for (int e = 0; e < G; ++e) {
                    if (e & half) continue;
                    const float2 wt = w[(1 << t) - 1 + (e >> (NS - t))];
                    const float wr = wt.x, wi = wt.y;
                    const f32x2 u = x[e], v = x[e + half], vs = v.yx;
                    x[e] = fma2(mk2(wr, wr), v, fma2(mk2(-wi, wi), vs, u));
                    x[e + half] = fma2(mk2(-wr, -wr), v, fma2(mk2(wi, -wi), vs, u));
                }
            }
#pragma unroll
            for (int e = 0; e < G; ++e) z[zslot(n0 + e * step)] = make_float2(x[e].x, x[e].y);
        }
    }
    wave_sync();
}

template <int LOG2W, int P, bool CACHED>
__device__ __forceinline__ void dit_all(float2* z, const float2* cache, const float* __restrict__ tw, int lane) {
    if constexpr (P == 0) {
        dit_pass0<LOG2W>(z, tw, lane);
        dit_all<LOG2W, 1, CACHED>(z, cache, tw, lane);
    } else if constexpr (P < Passes<LOG2W>::count) {
        dit_pass<LOG2W, P, CACHED>(z, cache, tw, lane);
        dit_all<LOG2W, P + 1, CACHED>(z, cache, tw, lane);
    }
}

// LDS floats per wave: N (+ skew) complex points, then one power term per bin the bands read
template <int LOG2W> constexpr int point_floats() { return 2 * ((1 << (LOG2W - 1)) + ((1 << (LOG2W - 1)) >> 5)); }
// (+ 7: the band sums read a partial batch of up to seven terms from a lane's own offset, up to six past kmax)
__host__ __device__ inline uint32_t bins_padded(uint32_t kmin, uint32_t kmax) { return (kmax - kmin + 7u + 63u) & ~63u; }

typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

template <int LOG2W, int WPB, bool CACHED>
__global__ __launch_bounds__(WPB * 64) void fft_bands_kernel(
    const void* __restrict__ pcm_raw, uint32_t fmt, uint64_t samples_per_clip, uint32_t stride,
    uint32_t windows_per_clip, uint64_t n_windows, const float* __restrict__ tw,
    const uint32_t* __restrict__ band_tbl, uint32_t nbands, uint32_t kmin, uint32_t kmax,
    float* __restrict__ frames) {
    constexpr int W = 1 << LOG2W;
    constexpr int N = W / 2;
    constexpr int LOGN = LOG2W - 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    constexpr int kZf = point_floats<LOG2W>();
    // LDS: [per-lane twiddle cache][split-pass twiddles W^k, k in [kmin, kmax)][per wave: points, power terms]
    float2* cache = reinterpret_cast<float2*>(smem);
    constexpr int kCacheFloats = CACHED ? 2 * Passes<LOG2W>::cache_slots : 0;
    const uint32_t nread = bins_padded(kmin, kmax);
    float2* split_tw = reinterpret_cast<float2*>(smem + kCacheFloats);
    float* zf = smem + kCacheFloats + 2 * nread + (size_t)wave * (kZf + nread);
    float2* z = reinterpret_cast<float2*>(zf);
    float* vbuf = zf + kZf;
    if constexpr (CACHED) build_cache<LOG2W, 0>(cache, tw, threadIdx.x, WPB * 64);
    for (uint32_t i = threadIdx.x; i < nread; i += WPB * 64) {
        const uint32_t k = kmin + i < (uint32_t)N ? kmin + i : 0u;
        split_tw[i] = make_float2(tw[k], tw[N + k]);
    }
    __syncthreads();   // the only workgroup barrier: the shared twiddle tables are complete
    const float inv_norm = 1.0f / (float)(W / 4);  // (Float32)(width/2), width = W/2; exact power of two

    // persistent waves: each walks the windows with a stride of the whole grid
    for (uint64_t win = (uint64_t)blockIdx.x * WPB + wave; win < n_windows; win += (uint64_t)gridDim.x * WPB) {
        const uint64_t clip = win / windows_per_clip;
        const uint32_t wi = (uint32_t)(win % windows_per_clip);
        const uint64_t first = clip * samples_per_clip + (uint64_t)wi * stride;
        // sample formats: 0 float32, 1 int16 / 32768, 2 int32 / 2^31 (what LBAudioDetectiveConvertToFormat,
        // LBAudioDetective.m:413-437, asks AudioConverter to do for integer PCM)
        // Sample i = lane + 64 q lands at float 2 zslot(i >> 1) + (i & 1) = lane + 66 q: each run of 64
        // samples is one contiguous LDS row, so float32 input goes straight from HBM/L2 to LDS
        // (global_load_lds_dword, no registers), and the integer formats need no per-sample index math.
        if (fmt == 0) {
#ifndef LBAD_EXP_NOLOAD
            const float* src = static_cast<const float*>(pcm_raw) + first + lane;
#pragma unroll 8
            for (int q = 0; q < W / 64; ++q)
                __builtin_amdgcn_global_load_lds((gvoid_t*)(src + 64 * q), (lvoid_t*)(zf + 66 * q), 4, 0, 0);
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
#endif
        } else if (fmt == 1) {
            const int16_t* src = static_cast<const int16_t*>(pcm_raw) + first + lane;
#pragma unroll 8
            for (int q = 0; q < W / 64; ++q) zf[lane + 66 * q] = (float)src[64 * q] * (1.0f / 32768.0f);
        } else {
            const int32_t* src = static_cast<const int32_t*>(pcm_raw) + first + lane;
#pragma unroll 8
            for (int q = 0; q < W / 64; ++q) zf[lane + 66 * q] = (float)src[64 * q] * (1.0f / 2147483648.0f);
        }
        wave_sync();

#ifndef LBAD_EXP_NOFFT
        dit_all<LOG2W, 0, CACHED>(z, cache, tw, lane);
#endif

        // split pass for the bins the bands read, then the reference's per-bin power term
        // (the trip count is wave-uniform -- nread is a multiple of 64 -- so two iterations can be in flight;
        // lanes past kmax redo bin kmin and drop the result)
#ifdef LBAD_EXP_NOSPLIT
        for (uint32_t it = 0; it < 1; ++it) {
#else
#pragma unroll 2
        for (uint32_t it = 0; it < nread / 64; ++it) {
#endif
            const uint32_t kk = kmin + lane + 64 * it;
            const bool live = kk < kmax;
            const uint32_t k = live ? kk : kmin;
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
                const float2 wk = split_tw[k - kmin];
                const float wr = wk.x, wi2 = wk.y;
                re = __fmaf_rn(wr, di, __fmaf_rn(wi2, dr, sr));
                im = __fmaf_rn(-wr, dr, __fmaf_rn(wi2, di, si));
            }
            if (re > 0.0f) re = __fmul_rn(re, inv_norm);
            if (im > 0.0f) im = __fmul_rn(im, inv_norm);
            const float pw = __fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im));
            // a NaN / inf term is skipped by the band sums (LBAudioDetective.m:398-401): +0 here leaves a sum of
            // squares -- never -0 -- unchanged, one select per bin instead of one per term read
            if (live) vbuf[k - kmin] = (pw == pw && fabsf(pw) != INFINITY) ? pw : 0.0f;
        }
        wave_sync();

        for (uint32_t b = lane; b < nbands; b += 64) {
#ifdef LBAD_EXP_NOBANDS
            const uint32_t lo = band_tbl[b], hi = lo + 1;
#else
            const uint32_t lo = band_tbl[b], hi = band_tbl[nbands + b];
#endif
            const float div = __uint_as_float(band_tbl[2 * nbands + b]);
            // the sum must run in bin order (float32 addition is not associative); the loads are issued
            // in batches of 8 so that their LDS latency overlaps instead of serialising with the adds.  Whole
            // batches need no per-term mask (a lane leaves the loop when its band runs out); the last w % 8
            // terms are read from the lane's own offset, what lies behind the band is selected away.
            float p = 0.0f;
            const uint32_t width = hi > lo ? hi - lo : 0;
            const float* vb = vbuf + (lo - kmin);
            const uint32_t full = width >> 3, rem = width & 7;
            for (uint32_t i = 0; i < full; ++i) {
                float x[8];
#pragma unroll
                for (uint32_t q = 0; q < 8; ++q) x[q] = vb[8 * i + q];
#pragma unroll
                for (uint32_t q = 0; q < 8; ++q) p = __fadd_rn(p, x[q]);
            }
            if (rem) {
                float x[7];
#pragma unroll
                for (uint32_t q = 0; q < 7; ++q) x[q] = vb[8 * full + q];
#pragma unroll
                for (uint32_t q = 0; q < 7; ++q) p = __fadd_rn(p, q < rem ? x[q] : 0.0f);
            }
            frames[win * nbands + b] = __fdiv_rn(p, div);
        }
        wave_sync();   // the next window overwrites the points and the power terms
    }
}

template <int LOG2W, int WPB, bool CACHED>
hipError_t launch_variant(const Plan& plan, const void* d_pcm, uint32_t fmt, uint64_t n_clips, uint64_t samples_per_clip,
                          uint32_t frames_per_clip, float* d_frames, hipStream_t stream) {
    constexpr size_t cache_bytes = CACHED ? (size_t)Passes<LOG2W>::cache_slots * sizeof(float2) : 0;
    const uint32_t nread = bins_padded(plan.table.kmin, plan.table.kmax);
    const size_t lds = cache_bytes + ((size_t)2 * nread + (size_t)WPB * (point_floats<LOG2W>() + nread)) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const uint32_t windows_per_clip = frames_per_clip * kRowsPerFrame;
    const uint64_t n_windows = n_clips * windows_per_clip;
    if (n_windows == 0) return hipSuccess;
    auto kern = fft_bands_kernel<LOG2W, WPB, CACHED>;
    static PerDevice attr;
    static int resident_on[kMaxDevices] = {};
    const int dev = current_device();
    if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
    if (attr.changed(lds + 1)) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, WPB * 64, lds) != hipSuccess || per_cu < 1)
            per_cu = 1;
        resident_on[dev] = device_cu_count() * per_cu;
    }
    const int resident = resident_on[dev];
    const uint64_t blocks_needed = (n_windows + WPB - 1) / WPB;
    const uint32_t grid = (uint32_t)(blocks_needed < (uint64_t)resident ? blocks_needed : (uint64_t)resident);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WPB * 64), lds, stream, d_pcm, fmt, samples_per_clip, plan.stride,
                       windows_per_clip, n_windows, plan.d_tw, plan.d_bands, plan.bands, plan.table.kmin,
                       plan.table.kmax, d_frames);
    return hipGetLastError();
}

// Waves per workgroup.  The per-lane twiddle cache is shared by the workgroup, every wave adds its own
// points and power terms, and waves never wait for each other, so the best shape is the largest
// workgroup that still fits one CU's 160 KB (one workgroup per CU, its waves spread over the SIMDs):
// W = 2048 -> 12 waves (3 per SIMD), W = 4096 -> 7.  Smaller windows keep 4 waves and several
// workgroups per CU.  plan.tune_waves / tune_cache (LBAudioDetectiveSetKernelTuning) override the choice for
// the LDS-tile sweep of BASELINE configs[4] (tools/sweep_lds_tiles.py).
template <int LOG2W, int WPB>
bool fits(const Plan& plan, bool cached) {
    const size_t cache_bytes = cached ? (size_t)Passes<LOG2W>::cache_slots * sizeof(float2) : 0;
    const uint32_t nread = bins_padded(plan.table.kmin, plan.table.kmax);
    return cache_bytes + ((size_t)2 * nread + (size_t)WPB * (point_floats<LOG2W>() + nread)) * sizeof(float) <= 160 * 1024;
}

template <int LOG2W>
hipError_t launch_one(const Plan& plan, const void* d_pcm, uint32_t fmt, uint64_t n_clips, uint64_t samples_per_clip,
                      uint32_t frames_per_clip, float* d_frames, hipStream_t stream) {
    const int want = (int)plan.tune_waves;
    const bool cached = plan.tune_cache;
#define LBAD_TRY(w)                                                                                              \
    if ((want == 0 || want == w) && fits<LOG2W, w>(plan, cached)) {                                             \
        if (cached) return launch_variant<LOG2W, w, true>(plan, d_pcm, fmt, n_clips, samples_per_clip, frames_per_clip, d_frames, stream); \
        return launch_variant<LOG2W, w, false>(plan, d_pcm, fmt, n_clips, samples_per_clip, frames_per_clip, d_frames, stream); \
    }
    if constexpr (LOG2W == 11) {
        LBAD_TRY(12) LBAD_TRY(8) LBAD_TRY(4) LBAD_TRY(2) LBAD_TRY(1)
    } else if constexpr (LOG2W == 12) {
        LBAD_TRY(7) LBAD_TRY(6) LBAD_TRY(4) LBAD_TRY(2) LBAD_TRY(1)
    } else if constexpr (LOG2W == 13) {
        LBAD_TRY(2) LBAD_TRY(1)
    } else {
        LBAD_TRY(4)
    }
#undef LBAD_TRY
    // the cache does not fit beside even the smallest workgroup: global twiddle loads
    if constexpr (LOG2W >= 12)
        return launch_variant<LOG2W, 1, false>(plan, d_pcm, fmt, n_clips, samples_per_clip, frames_per_clip, d_frames, stream);
    return launch_variant<LOG2W, 4, false>(plan, d_pcm, fmt, n_clips, samples_per_clip, frames_per_clip, d_frames, stream);
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
