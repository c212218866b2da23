// k_rows_pruned.hip -- specialised stage 1 for 1024-sample windows whose bands read only the
// lowest FFT bins (44.1 kHz / 1024: bins 0..21): windows -> 128 x 32 frame rows.
//
// Same arithmetic as k_fft_bands.hip / oracle rfft_exec (radix-2 DIT, nested-fma butterflies),
// but only the butterflies the 22 consumed bins depend on are executed:
//
//   * 8 lanes own one window.  Lane r holds the 64 complex points z[r + 8m] and runs DIT stages
//     1..5 (two 32-point blocks) entirely in registers with compile-time twiddles.
//   * Stage 6 is pruned to the 43 outputs k64 in [0,21] u [43,63] that feed bins 0..21 and their
//     mirror bins 491..511 (needed by the real-FFT split pass).
//   * Stages 7..9 combine the 8 lanes.  Each needed bin is a 7-butterfly reduction tree over the
//     8 lanes' values of one stage-6 output; the values cross lanes through a per-wave LDS
//     transpose buffer and one lane evaluates both trees of a bin pair (k, 512-k), the split
//     pass, the positive-only normalisation and the power term.
//   * The work unit is a quarter frame (32 windows): its PCM span (31 * 64 + 1024 samples) goes from
//     HBM to LDS once, so the 16x window overlap costs no extra HBM traffic.  Persistent workgroups
//     (two per CU) claim frames from per-XCD counters and refill the span buffer with
//     global_load_lds behind the arithmetic of the current unit (see frame_rows_pruned_kernel).
//
// Executed work per window: ~10.3 k float operations instead of 25.6 k for the full transform;
// results are bit-identical because every surviving butterfly is evaluated exactly as in the
// full network.
#include "internal.hpp"
#include "fft64_lane.hpp"

namespace lbad {
namespace {

constexpr int kW = 1024;
constexpr int kStride = 64;
constexpr int kBands = 32;
constexpr int kBins = 22;           // bins 0..21
constexpr int kRows = 43;           // stage-6 outputs kept per lane
constexpr int kRowsA = 22;          // pass 1: rows 0..21 ("+" side), pass 2: rows 22..42 (mirror side)
constexpr int kRowDw = 20;          // dwords per transpose row: 8 lanes x 8 B, padded 64 -> 80 B
constexpr int kWinDw = 464;         // 22 rows x 20 = 440, padded so that window stride = 16 (mod 32) banks
constexpr int kWaves = 4;
constexpr int kThreads = kWaves * 64;
constexpr int kUnitWindows = 32;    // a workgroup owns a quarter frame: 4 waves x 8 windows
constexpr int kSpan = (kUnitWindows - 1) * kStride + kW;  // 3008 samples
constexpr int kSpanDw = kSpan + 16 * (kSpan >> 6);        // 16-dword skew per 64 samples: 3760
constexpr int kTDw = kWaves * 8 * kWinDw;                 // per wave: 8 windows x one pass of rows
constexpr int kBinConst = 14;       // per-bin twiddle block, see rows_pruned_constants()
constexpr int kConstStride = 20;    // 14 floats per bin padded to 80 B: lanes with different bins hit different banks
constexpr int kConstDw = kBins * kConstStride;
constexpr int kClaimOffset = 312;   // plan.d_bin_const: 308 constants, padding, then 8 claim counters (one per XCD)
constexpr int kLdsBytes = (kSpanDw + kTDw + kConstDw + 4) * 4;   // 75 856 B: two workgroups per CU
constexpr int kWgPerCu = 2;
static_assert(kWgPerCu * kLdsBytes <= 160 * 1024, "two workgroups must fit one CU's LDS");

using namespace lane64;

// stage-6 outputs: row i < 22 is k64 = i ("+" output of pair j = i); row i >= 22 is k64 = i + 21
// ("-" output of pair j = i - 11)
template <int I>
__device__ __forceinline__ cplx stage6_row(const cplx (&x)[64]) {
    constexpr int j = I < 22 ? I : I - 11;
    constexpr bool plus = I < 22;
    const cplx u = x[j], v = x[j + 32];
    if constexpr (j == 0) {
        return u + v;   // only the "+" output of pair 0 is needed
    } else if constexpr (j == 16) {
        return plus ? fma2(mk(1.0f, -1.0f), v.yx, u) : fma2(mk(-1.0f, 1.0f), v.yx, u);
    } else {
        constexpr float wr = kTw64Re[j], wi = kTw64Im[j];
        if constexpr (plus) return fma2(mk(wr, wr), v, fma2(mk(-wi, wi), v.yx, u));
        else return fma2(mk(-wr, -wr), v, fma2(mk(wi, -wi), v.yx, u));
    }
}

// rows [I, END) of stage 6 -> transpose buffer rows [I - FIRST, ...).  volatile keeps the compiler
// from fusing pairs into ds_write2_b64 / ds_read2_b64, which run at half the rate of the plain
// 64-bit forms on gfx950.
template <int I, int END, int FIRST>
__device__ __forceinline__ void store_rows(const cplx (&x)[64], float* trow) {
    if constexpr (I < END) {
        *(lds_vf32x2*)(trow + (I - FIRST) * kRowDw) = stage6_row<I>(x);
        store_rows<I + 1, END, FIRST>(x, trow);
    }
}

template <int I, int END, int FIRST>
__device__ __forceinline__ void hold_rows(const cplx (&x)[64], cplx (&h)[kRows - kRowsA]) {
    if constexpr (I < END) {
        h[I - FIRST] = stage6_row<I>(x);
        hold_rows<I + 1, END, FIRST>(x, h);
    }
}
template <int I, int N>
__device__ __forceinline__ void store_held(const cplx (&h)[kRows - kRowsA], float* trow) {
    if constexpr (I < N) {
        *(lds_vf32x2*)(trow + I * kRowDw) = h[I];
        store_held<I + 1, N>(h, trow);
    }
}

template <int M>
__device__ __forceinline__ void load_points(cplx (&x)[64], const float* src) {
    if constexpr (M < 64) {
        // register slot I holds point m = brev6(I): sample 2r + 16 m of the window sits
        // (m & 3) * 16 + (m >> 2) * 80 dwords after the lane base.  Loading in slot order lets the
        // first butterflies start while the later points are still in flight.
        constexpr int m = brev6(M);
        x[M] = *(const lds_vf32x2*)(src + (m & 3) * 16 + (m >> 2) * 80);
        load_points<M + 1>(x, src);
    }
}

// 7-butterfly reduction over the 8 lanes' values of one stage-6 row (DIT stages 7, 8, 9).  Operands
// and evaluation are separate so that a pass can have the reads of all its rounds in flight at once.
struct TreeIn {
    float4 q0, q1, q2, q3;
};
struct TreeTw {     // twiddles of one tree: W_128^k, W_256^k, W_512^k (or their mirror forms)
    f32x2 t0, t1, t2;
};
__device__ __forceinline__ TreeIn tree_load(const float* row) {
    TreeIn in;
    in.q0 = *reinterpret_cast<const float4*>(row);
    in.q1 = *reinterpret_cast<const float4*>(row + 4);
    in.q2 = *reinterpret_cast<const float4*>(row + 8);
    in.q3 = *reinterpret_cast<const float4*>(row + 12);
    return in;
}
__device__ __forceinline__ TreeTw tree_twiddles(const float* tw) {
    TreeTw t;
    t.t0 = *reinterpret_cast<const f32x2*>(tw);
    t.t1 = *reinterpret_cast<const f32x2*>(tw + 2);
    t.t2 = *reinterpret_cast<const f32x2*>(tw + 4);
    return t;
}
__device__ __forceinline__ cplx tree_eval(const TreeIn& in, const TreeTw& t) {
    const cplx x0 = mk(in.q0.x, in.q0.y), x1 = mk(in.q0.z, in.q0.w), x2 = mk(in.q1.x, in.q1.y), x3 = mk(in.q1.z, in.q1.w);
    const cplx x4 = mk(in.q2.x, in.q2.y), x5 = mk(in.q2.z, in.q2.w), x6 = mk(in.q3.x, in.q3.y), x7 = mk(in.q3.z, in.q3.w);
    const cplx y0 = madd(x0, t.t0.x, t.t0.y, x4), y1 = madd(x1, t.t0.x, t.t0.y, x5);
    const cplx y2 = madd(x2, t.t0.x, t.t0.y, x6), y3 = madd(x3, t.t0.x, t.t0.y, x7);
    const cplx z0 = madd(y0, t.t1.x, t.t1.y, y2), z1 = madd(y1, t.t1.x, t.t1.y, y3);
    return madd(z0, t.t2.x, t.t2.y, z1);
}

// per-bin constants (kBinConst floats, padded to kConstStride): [0..5] twiddles of the "+" tree (stages 7, 8,
// 9), [6..11] of the mirror tree, [12..13] split-pass twiddle W_1024^k
//
// The span of the quarter frame starting at sample `first` goes to LDS skewed by 16 dwords per 64
// samples (so that the 4 windows of a 32-lane group hit disjoint bank quarters when the points are read).
// FMT 0 (float32): one global_load_lds_dword per 64 samples -- the data never passes through VGPRs
// and the wave does not wait for it; FMT 1 / 2 (int16 / 32768, int32 / 2^31) convert in registers.
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

template <int FMT>
__device__ __forceinline__ void span_to_lds(const void* __restrict__ pcm_raw, uint64_t first, float* span, int wave,
                                            int lane) {
    if constexpr (FMT == 0) {
        // `wave` is wave-uniform (an SGPR): chunk c = wave + 4 i, fully unrolled, LDS base through M0
        const float* src = static_cast<const float*>(pcm_raw) + first + lane + 64 * wave;
        float* dst = span + 80 * wave;
        constexpr int kFull = (kSpan / 64) / kWaves;      // 47 chunks: 11 rounds of 4, then waves 0..2
#pragma unroll
        for (int i = 0; i < kFull; ++i)
            __builtin_amdgcn_global_load_lds((gvoid_t*)(src + 64 * kWaves * i), (lvoid_t*)(dst + 80 * kWaves * i), 4, 0, 0);
        if (wave < kSpan / 64 - kFull * kWaves)
            __builtin_amdgcn_global_load_lds((gvoid_t*)(src + 64 * kWaves * kFull), (lvoid_t*)(dst + 80 * kWaves * kFull), 4, 0, 0);
    } else if constexpr (FMT == 1) {
        const int16_t* src = static_cast<const int16_t*>(pcm_raw) + first;
        for (int s = threadIdx.x; s < kSpan; s += kThreads)
            span[s + 16 * (s >> 6)] = (float)src[s] * (1.0f / 32768.0f);
    } else {
        const int32_t* src = static_cast<const int32_t*>(pcm_raw) + first;
        for (int s = threadIdx.x; s < kSpan; s += kThreads)
            span[s + 16 * (s >> 6)] = (float)src[s] * (1.0f / 2147483648.0f);
    }
}

// Persistent workgroups, two per CU.  Every XCD owns a contiguous range of frames; its workgroups take
// the first ones statically and claim the rest, a frame (four quarter frames) at a time, from a per-XCD
// counter (the waves of a workgroup do not land evenly on the SIMDs, so equal static shares finish up
// to 40 % apart).  As soon as every wave holds its points of quarter frame u in registers, the span of
// the next one streams into the same LDS buffer behind the arithmetic; the claim for the frame after
// is in flight for three iterations, and the rows of u drain to HBM during u + 1.
template <int FMT>
__global__ __launch_bounds__(kThreads, kWgPerCu) void frame_rows_pruned_kernel(const void* __restrict__ pcm_raw,
                                                                         uint64_t samples_per_clip,
                                                                         uint32_t frames_per_clip, uint64_t n_units,
                                                                         uint64_t units_per_xcd,
                                                                         const float* __restrict__ bin_const,
                                                                         const uint32_t* __restrict__ band_tbl,
                                                                         uint32_t* __restrict__ claim_ctr,
                                                                         float* __restrict__ frames, uint32_t frame_dw,
                                                                         uint32_t place_tbl) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* span = smem;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    float* tbuf = smem + kSpanDw + wave * (8 * kWinDw);
    float* vbuf = tbuf;                       // power terms reuse the wave's transpose area after the trees
    float* cbuf = smem + kSpanDw + kTDw;
    uint32_t* claim_slot = reinterpret_cast<uint32_t*>(cbuf + kConstDw);

    // Workgroup b runs on XCD b % 8 (observed dispatch order; used for speed only: a workgroup on
    // another XCD would see its own copy of the counter and repeat work, not skip any).  Giving every
    // XCD a contiguous range makes neighbouring quarter frames -- whose PCM spans overlap by a third --
    // share one L2 instead of fetching the overlap from HBM once per XCD.
    const uint32_t wg_per_xcd = gridDim.x >> 3;
    const uint64_t xcd_begin = (uint64_t)(blockIdx.x & 7) * units_per_xcd;     // units_per_xcd is a multiple of 4
    const uint64_t xcd_end = xcd_begin + units_per_xcd < n_units ? xcd_begin + units_per_xcd : n_units;
    uint64_t unit = xcd_begin + 4 * (uint64_t)(blockIdx.x >> 3);
    if (unit >= xcd_end) return;

    auto span_start = [&](uint64_t u) {
        const uint32_t frame = (uint32_t)(u >> 2);          // the launcher keeps unit numbers below 2^31
        const uint32_t clip = frame / frames_per_clip;
        const uint32_t fi = frame - clip * frames_per_clip;
        return (uint64_t)clip * samples_per_clip + (uint64_t)(fi * 128 + (uint32_t)(u & 3) * kUnitWindows) * kStride;
    };
    uint32_t* my_ctr = claim_ctr + (blockIdx.x & 7);
    const bool claimer = threadIdx.x == 0;
    uint32_t claimed = 0;

    span_to_lds<FMT>(pcm_raw, span_start(unit), span, wave, lane);
    for (int i = threadIdx.x; i < kBins * kBinConst; i += kThreads)
        cbuf[(i / kBinConst) * kConstStride + (i % kBinConst)] = bin_const[i];
    const int band = lane & 31;
    const uint32_t b_lo = band_tbl[band], b_hi = band_tbl[kBands + band];
    const float b_div = __uint_as_float(band_tbl[2 * kBands + band]);
    // where the band's mean of row w goes inside a frame of frame_dw floats: w * b_mult + b_off -- rows of 32 bands, or
    // the compact frame of plan.sparse (128 rows of the bands that can be non-zero; b_off = 0xFFFFFFFF: a band that is +0.0
    // in every window and that nobody reads)
    const uint32_t b_mult = band_tbl[place_tbl * kBands + band], b_off = band_tbl[(place_tbl + 1) * kBands + band];

    const int w8 = lane >> 3, r = lane & 7;
    const float inv_norm = 1.0f / (float)(kW / 4);
    // bin tasks: the 8 lanes of a window share its 22 bins, lane r taking bins r', r' + 8, r' + 16 (< 22): eight lanes
    // read eight consecutive rows of one window.  r' = r, or r ^ 4 in the windows 1, 2 (mod 4) of the wave (round 5):
    // a ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... -- lanes 0-3 of window
    // 0, 4-7 of windows 1 and 2, 0-3 of window 3 --, and with r' = r throughout two of the four windows of a group
    // met on the same 16-byte slots (window pitch 116 slots = 4 mod 16, rows 5 slots apart): every tree read took twice
    // its cycles, 18 % of the kernel's LDS-active cycles were conflicts.  With the halves of windows 1 and 2 swapped the
    // sixteen lanes of every group read sixteen different slots (the pitch stays: the 16 contiguous lanes of a
    // ds_write_b64 group want the windows 16 banks apart).
#ifdef LBAD_EXP_TREE_NOSWAP
    const int rr = r;
#else
    const int rr = r ^ ((((w8 + 1) >> 1) & 1) << 2);
#endif
    int tk[3];
#pragma unroll
    for (int rd = 0; rd < 3; ++rd) tk[rd] = rr + 8 * rd < kBins ? rr + 8 * rd : kBins - 1;
    const bool last_valid = rr + 16 < kBins;

    // a lane serves the same three (window, bin) tasks in every quarter frame: their twiddles stay in registers
    __syncthreads();
    TreeTw twp[3], twm[3];
    f32x2 ws[3];
#pragma unroll
    for (int rd = 0; rd < 3; ++rd) {
        const float* c = cbuf + tk[rd] * kConstStride;
        twp[rd] = tree_twiddles(c);
        twm[rd] = tree_twiddles(c + 6);
        ws[rd] = *reinterpret_cast<const f32x2*>(c + 12);
    }

    float out[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // band means of the previous quarter frame, stored one iteration late
    float* out_ptr = nullptr;
#ifdef LBAD_EXP_TIMELINE
    long long ts[8];
#define STAMP(i) ts[i] = __builtin_readcyclecounter()
#else
#define STAMP(i)
#endif
    for (;;) {
        STAMP(0);
#ifdef LBAD_EXP_TIMELINE
        const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
        // ---- A: this quarter frame's span has landed (own loads: vmcnt, the other waves': barrier) ----
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        const uint32_t quarter = (uint32_t)(unit & 3);
        if (quarter == 3 && claimer) claim_slot[0] = claimed;    // claimed three iterations ago
        __syncthreads();
        STAMP(1);

        // ---- B1: 64 points of this lane; once every wave has its points the span buffer is free -------
        cplx x[64];
        load_points<0>(x, span + 80 * (8 * wave + w8) + 2 * r);
        const uint64_t next = quarter == 3 ? xcd_begin + 4 * ((uint64_t)wg_per_xcd + claim_slot[0]) : unit + 1;
        __syncthreads();
        if (next < xcd_end) span_to_lds<FMT>(pcm_raw, span_start(next), span, wave, lane);
        if (quarter == 0 && claimer) claimed = atomicAdd(my_ctr, 1u);
#ifndef LBAD_EXP_TIMELINE
        if (out_ptr) {
#pragma unroll
            for (int q = 0; q < 4; ++q) out_ptr[q * 2 * b_mult] = out[q];
        }
#endif
        STAMP(2);

        // ---- B2: DIT stages 1..5 in registers ---------------------------------------------------------
        // Issue priority: low while this wave is the arithmetic-heavy one, high for everything else (LDS
        // transposes, point reads, prefetch issue), whose short instructions should not queue behind the
        // co-resident wave's butterflies (17.3 -> 16.9 ms).
        __builtin_amdgcn_s_setprio(0);
        stage_blocks<1, 0>(x);
        stage_blocks<2, 0>(x);
        stage_blocks<3, 0>(x);
        stage_blocks<4, 0>(x);
        stage_blocks<5, 0>(x);
        STAMP(3);
        __builtin_amdgcn_s_setprio(3);

        // ---- B3/B4: stage 6, then the "+" rows through the transpose buffer; the mirror rows wait in
        //      registers (the 64 points are dead from here on, which leaves room to keep every read of a
        //      pass in flight) -----------------------------------------------------------------------------
        store_rows<0, kRowsA, 0>(x, tbuf + w8 * kWinDw + 2 * r);
        cplx held[kRows - kRowsA];
        hold_rows<kRowsA, kRows, kRowsA>(x, held);
        cplx za[3];
        {
            TreeIn in[3];
#pragma unroll
            for (int rd = 0; rd < 3; ++rd) in[rd] = tree_load(tbuf + w8 * kWinDw + tk[rd] * kRowDw);
            // ---- pass 2: the mirror rows reuse the same buffer.  LDS executes a wave's operations in
            //      order, so these stores cannot pass the reads above, and they are on their way while
            //      the trees of pass 1 are evaluated. ------------------------------------------------------
            store_held<0, kRows - kRowsA>(held, tbuf + w8 * kWinDw + 2 * r);
#pragma unroll
            for (int rd = 0; rd < 3; ++rd) za[rd] = tree_eval(in[rd], twp[rd]);
        }
        STAMP(4);
        float pw[3];
        {
            TreeIn in[3];
#pragma unroll
            for (int rd = 0; rd < 3; ++rd) {
                const int k = tk[rd];
                // mirror bin 512 - k is stage-6 row 43 - k, i.e. row 21 - k of this pass (bin 0 has no mirror:
                // its lane reads the stale row 21, which keeps the stride, and drops the result)
                in[rd] = tree_load(tbuf + w8 * kWinDw + (21 - k) * kRowDw);
            }
#pragma unroll
            for (int rd = 0; rd < 3; ++rd) {
                const int k = tk[rd];
                const cplx b = tree_eval(in[rd], twm[rd]);
                const cplx a = za[rd];
                float re, im;
                if (k == 0) {
                    const float sm = a.x + a.y, df = a.x - a.y;
                    re = sm + sm;
                    im = df + df;
                } else {
                    const float sr = a.x + b.x, si = a.y - b.y;
                    const float dr = a.x - b.x, di = a.y + b.y;
                    const float wr = ws[rd].x, wi = ws[rd].y;
                    re = __fmaf_rn(wr, di, __fmaf_rn(wi, dr, sr));
                    im = __fmaf_rn(-wr, dr, __fmaf_rn(wi, di, si));
                }
                if (re > 0.0f) re = __fmul_rn(re, inv_norm);
                if (im > 0.0f) im = __fmul_rn(im, inv_norm);
                pw[rd] = __fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im));
            }
        }
        STAMP(5);
        // power terms -> LDS only after every tree of the wave has read its rows
#pragma unroll
        for (int rd = 0; rd < 3; ++rd)
            if (rd < 2 || last_valid) vbuf[w8 * 24 + tk[rd]] = pw[rd];

        // ---- B5: band means, 8 windows x 32 bands per wave.  The four windows a lane serves advance
        //      together: one LDS round trip per bin instead of four. -----------------------------------
        float p[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        const float* vb = vbuf + (lane >> 5) * 24;
        for (uint32_t k = b_lo; k < b_hi; ++k) {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = vb[q * 48 + k];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (v[q] == v[q] && fabsf(v[q]) != INFINITY) p[q] = __fadd_rn(p[q], v[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) out[q] = __fdiv_rn(p[q], b_div);
        // row = quarter * 32 + 8 * wave + 2 q + (lane >> 5)
        out_ptr = b_off != 0xFFFFFFFFu
                      ? frames + (unit >> 2) * frame_dw + (quarter * kUnitWindows + 8 * wave + (lane >> 5)) * b_mult + b_off
                      : nullptr;
#ifdef LBAD_EXP_TIMELINE
        STAMP(6);
        if (lane == 0) {
            float* o = frames + ((unit >> 2) * 128 + quarter * kUnitWindows + 8 * wave) * kBands;
            for (int i = 1; i < 7; ++i) o[i - 1] = (float)(ts[i] - ts[i - 1]);
            o[6] = (float)(ts[0] & 0xFFFFFF);
            o[7] = (float)__builtin_amdgcn_s_getreg(((16 - 1) << 11) | 4);
            o[8] = (float)(ts[0] >> 24 & 0xFFFFFF);
            o[9] = (float)(__builtin_amdgcn_s_memrealtime() - rt0);
        }
#endif
        if (next >= xcd_end) break;
        unit = next;
    }
#ifndef LBAD_EXP_TIMELINE
    if (out_ptr) {
#pragma unroll
        for (int q = 0; q < 4; ++q) out_ptr[q * 2 * b_mult] = out[q];
    }
#endif
}

}  // namespace

bool rows_pruned_supported(const Plan& p) {
    if (p.window != (uint32_t)kW || p.stride != (uint32_t)kStride || p.bands != (uint32_t)kBands) return false;
    if (p.table.kmax > (uint32_t)kBins) return false;
    // the compile-time W_64 table must be bit-identical to the run-time master table
    std::vector<float> re, im;
    make_twiddles(kW, re, im);
    for (int t = 0; t < 32; ++t)
        if (re[16 * t] != kTw64Re[t] || im[16 * t] != kTw64Im[t]) return false;
    return true;
}

// per-bin twiddle block (see kBinConst)
void rows_pruned_constants(std::vector<float>& out) {
    std::vector<float> re, im;
    make_twiddles(kW, re, im);
    out.assign((size_t)kClaimOffset + 8, 0.0f);   // constants, padding, claim counters
    for (int k = 0; k < kBins; ++k) {
        float* c = &out[(size_t)k * kBinConst];
        // "+" tree of Z[k]: W_128^k, W_256^k, W_512^k
        c[0] = re[8 * k]; c[1] = im[8 * k];
        c[2] = re[4 * k]; c[3] = im[4 * k];
        c[4] = re[2 * k]; c[5] = im[2 * k];
        if (k > 0) {
            // mirror tree of Z[512 - k]: exponents 128 - k, 256 - k, 512 - k all lie in the upper half of
            // their stage, i.e. the "u - w v" output: w' = -W^(e - m/2)
            c[6] = -re[8 * (64 - k)];   c[7] = -im[8 * (64 - k)];
            c[8] = -re[4 * (128 - k)];  c[9] = -im[4 * (128 - k)];
            c[10] = -re[2 * (256 - k)]; c[11] = -im[2 * (256 - k)];
            c[12] = re[k]; c[13] = im[k];   // W_1024^k of the split pass
        }
    }
}

template <int FMT>
static hipError_t launch_rows_fmt(const Plan& plan, const float* d_bin_const, const void* d_pcm, uint64_t n_frames,
                                  uint64_t samples_per_clip, uint32_t frames_per_clip, float* d_frames,
                                  hipStream_t stream, bool compact) {
    static PerDevice attr;
    if (attr.changed(kLdsBytes)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(frame_rows_pruned_kernel<FMT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        if (e != hipSuccess) return e;
    }
    // work is claimed by frames: every XCD's range is a whole number of frames
    const uint64_t n_units = n_frames * 4, units_per_xcd = 4 * ((n_frames + 7) / 8);
    const int n_cu = device_cu_count();
    // every CU holds kWgPerCu persistent workgroups; a multiple of 8 so that each XCD gets the same number
    uint64_t wg_per_xcd = ((uint64_t)n_cu * kWgPerCu + 7) / 8;
    if (wg_per_xcd > units_per_xcd / 4) wg_per_xcd = units_per_xcd / 4;
    uint32_t* claim = reinterpret_cast<uint32_t*>(const_cast<float*>(d_bin_const)) + kClaimOffset;
    hipError_t e = hipMemsetAsync(claim, 0, 8 * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(frame_rows_pruned_kernel<FMT>, dim3((uint32_t)(wg_per_xcd * 8)), dim3(kThreads), kLdsBytes,
                       stream, d_pcm, samples_per_clip, frames_per_clip, n_units, units_per_xcd, d_bin_const,
                       plan.d_bands, claim, d_frames, compact ? plan.sparse.frame_dw() : 128u * kBands, compact ? 5u : 3u);
    return hipGetLastError();
}

hipError_t launch_rows_pruned(const Plan& plan, const float* d_bin_const, const void* d_pcm, uint32_t fmt,
                              uint64_t n_clips, uint64_t samples_per_clip, uint32_t frames_per_clip, float* d_frames,
                              hipStream_t stream, bool compact) {
    if (compact && !plan.sparse.ok) return hipErrorInvalidValue;
    const uint64_t n_frames = n_clips * frames_per_clip;
    if (n_frames == 0) return hipSuccess;
    if (n_frames > 0x7fffffffull) return hipErrorInvalidValue;
    if (n_frames * 4 + 8 > 0x7fffffffull) return hipErrorInvalidValue;
    switch (fmt) {
        case 0: return launch_rows_fmt<0>(plan, d_bin_const, d_pcm, n_frames, samples_per_clip, frames_per_clip, d_frames, stream, compact);
        case 1: return launch_rows_fmt<1>(plan, d_bin_const, d_pcm, n_frames, samples_per_clip, frames_per_clip, d_frames, stream, compact);
        case 2: return launch_rows_fmt<2>(plan, d_bin_const, d_pcm, n_frames, samples_per_clip, frames_per_clip, d_frames, stream, compact);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace lbad
