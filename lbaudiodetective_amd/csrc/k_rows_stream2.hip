// k_rows_stream2.hip -- streaming stage 1 for 2048-sample windows at stride 64 (the reference's DEFAULT
// configuration, 5512 Hz / 2048, BASELINE configs[0]): windows -> 128 x bands frame rows.
//
// The plan of k_rows_stream.hip one size down (N = 1024 complex points, hop 32): the stage-4 sub-transforms
//     D4(g) = DFT16 { c[g + 64 m] },  g = 32 i + n,  n in [0, 64)
// of window i are bit for bit window (i + 1)'s for n >= 32, so a lane that walks a clip in time transforms
// 16 new points per window, keeps the previous D4 in registers and forms E(g) = D4(g) +- W_32^k D4(g + 32)
// (stage 5, a 32-point transform over stride 32) itself -- no exchange between lanes.  32 lanes make a
// window, so a wave walks TWO runs of windows at once (lanes 0..31 / 32..63), and the 2 x 32 rows of E they
// emit per step are exactly one row per lane for phase 2:
//
//   phase 1  lane (n, s): 16 points from L2 (issued a window ahead) -> D4 (DIT stages 1..4 in registers) ->
//            stage 5 with the previous step's D4, compile-time twiddles -> rows k and k + 16 of the window's
//            32 x 32 matrix E into the wave's LDS transpose.
//   phase 2  lane = one row k32 of one of the two windows: the 32-point cross transform over n (stages 6..10,
//            per-row twiddles from LDS) in full -- the default band table reads bins 86..758 of 1024, their
//            mirrors included every output is needed --, partner rows (a, 32 - a) in neighbouring lanes trade
//            outputs by DPP for the split pass; power terms -> LDS, band sums in bin order, lanes 0..31 for
//            the first window, 32..63 for the second.
//
// Executed butterflies per window: 32 x (32 + 16) + 32 x 80 = 4.1 k against 5.1 k for the full transform and
// 798 VALU wave-instructions per window in k_rows_full.hip against ~500 here.
#include "stream_common.hpp"

namespace lbad {
namespace {

using namespace lane64;
using namespace stream;

constexpr int kW = 2048;
constexpr int kN = kW / 2;
constexpr int kStride = 64;
constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;
constexpr int kChunk = 16;                    // windows per run (see k_rows_stream.hip: keeps the sliding windows in L2)
constexpr int kChunksPerFrame = 128 / kChunk;
constexpr int kRowDw = 68;                    // 32 complex + 16 B: conflict-free ds_read_b128 across lanes
constexpr int kTDw = 64 * kRowDw;             // two windows x 32 rows
constexpr int kCrossTw = 31;                  // 1 + 2 + 4 + 8 + 16 twiddles of a row's cross transform
constexpr int kCtwDw = 66;                    // pitch of a row's twiddles: 32 lanes x ds_read_b64 land on 64 different banks
                                              // (round 5; at the transpose's pitch of 68 lanes n and n + 16 shared theirs)
constexpr int kMaxTerms = 56;                 // bins of the widest band (the band sums are unrolled this far)
// Power terms of a window (round 5): NOT by bin number -- lane b of a half-wave adds band b's terms in bin order, and with
// the bands' first bins anywhere the 32 lanes of a ds_read_b32 met on the same banks four at a time (189 extra LDS cycles
// on 63 reads per window with the default table: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 45.7 %, most of it here).  The
// bands' terms now lie one band after the other, every band starting on a bank (word mod 32) of its own (BandTable::
// term_at, made by the host: 58 words of gaps for the default table): term j of every band is on a different bank.  The
// lanes that PRODUCE the terms (row lane n: bins row + 32 q) take each bin's word from a table in LDS; a store of 32
// consecutive bins then conflicts only where a band ends (11 extra cycles on 22 stores); bins no band reads go to a dump
// word per lane behind the last band.
constexpr int kPowerDw = kTDw / 2;            // one window's area: <= 1024 terms + 32 x 31 gap words + the read overrun
static_assert(kN + 32 * 31 + kMaxTerms + 7 <= kPowerDw, "skewed power terms of a window fit its half of the transpose area");
constexpr int kLdsDw = kWaves * kTDw + 32 * kCtwDw + 32 * 32 * 2 + 32 * 32;
constexpr int kLdsBytes = kLdsDw * 4;         // 159 744 B (+ 384 B static): one workgroup per CU
static_assert(kLdsBytes + 512 <= 160 * 1024, "LDS budget");

// row of lane l (within its window): lanes 2 p, 2 p + 1 hold the partner rows (p, 32 - p); pair 0 is (0, 16)
__device__ __forceinline__ int row_of_lane(int l) {
    const int p = l >> 1;
    return (l & 1) == 0 ? p : (p == 0 ? 16 : 32 - p);
}

// stage 5: E[K] = u + W_32^K v, E[K + 16] = u - W_32^K v with the oracle's special cases (K = 0, K = 8)
template <int K>
__device__ __forceinline__ void stage5(const cplx u, const cplx v, cplx& ep, cplx& em) {
    if constexpr (K == 0) {
        ep = u + v;
        em = u - v;
    } else if constexpr (K == 8) {        // w = -i
        ep = fma2(mk(1.0f, -1.0f), v.yx, u);
        em = fma2(mk(-1.0f, 1.0f), v.yx, u);
    } else {
        constexpr float wr = kTw64Re[2 * K], wi = kTw64Im[2 * K];
        ep = fma2(mk(wr, wr), v, fma2(mk(-wi, wi), v.yx, u));
        em = fma2(mk(-wr, -wr), v, fma2(mk(wi, -wi), v.yx, u));
    }
}

template <int K>
__device__ __forceinline__ void emit_rows(const cplx (&P)[16], const cplx (&Nn)[16], float* col) {
    if constexpr (K < 16) {
        cplx ep, em;
        stage5<K>(P[K], Nn[K], ep, em);
        // row k -> lane 2 k (row 0 -> lane 0), row 16 + k -> lane 33 - 2 k (row 16 -> lane 1)
        *(lds_vf32x2*)(col + 2 * K * kRowDw) = ep;
        *(lds_vf32x2*)(col + (K == 0 ? 1 : 33 - 2 * K) * kRowDw) = em;
        emit_rows<K + 1>(P, Nn, col);
    }
}

// QLO, QHI: the q (bins row + 32 q) some band reads, compile time so that the split pass is straight-line code
template <int FMT, int QLO, int QHI>
__global__ __launch_bounds__(kThreads, 2) void rows_stream2_kernel(const void* __restrict__ pcm, uint64_t samples_per_clip,
                                                                   uint32_t frames_per_clip, uint32_t n_runs,
                                                                   uint32_t runs_per_xcd, const float* __restrict__ tw,
                                                                   const uint32_t* __restrict__ band_tbl, uint32_t nbands,
                                                                   uint32_t kmin, uint32_t kmax, uint32_t n_batches,
                                                                   uint32_t* __restrict__ claim_ctr,
                                                                   float* __restrict__ frames) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    float* tbuf = smem + wave * kTDw;                       // this wave's transpose / power terms
    float* ctw = smem + kWaves * kTDw;                      // [row lane 0..31][31 complex], pitch kCtwDw
    float2* stw = reinterpret_cast<float2*>(ctw + 32 * kCtwDw);   // [q][row lane]
    uint32_t* term_at = reinterpret_cast<uint32_t*>(stw + 32 * 32);   // [q][row lane]: word of bin row + 32 q in the window's area
    __shared__ uint32_t s_edge[3][32];                      // the bands' first bins, ends and first words

    // ---- once per workgroup: tables --------------------------------------------------------------------
    for (int i = threadIdx.x; i < 32 * kCrossTw; i += kThreads) {
        const int l = i / kCrossTw, e = i % kCrossTw;
        const int a = row_of_lane(l);
        int s = 1;
        while ((1 << s) - 1 <= e) ++s;                      // cross stage s = 1..5 (overall stage 5 + s)
        const int jj = e - ((1 << (s - 1)) - 1);
        const uint32_t ti = (uint32_t)(a + 32 * jj) << (6 - s);     // W_(32 * 2^s)^(a + 32 jj)
        ctw[l * kCtwDw + 2 * e] = tw[ti];
        ctw[l * kCtwDw + 2 * e + 1] = tw[kN + ti];
    }
    if (threadIdx.x < 32) {
        const bool live = threadIdx.x < nbands;
        s_edge[0][threadIdx.x] = live ? band_tbl[threadIdx.x] : 0u;
        s_edge[1][threadIdx.x] = live ? band_tbl[nbands + threadIdx.x] : 0u;
        s_edge[2][threadIdx.x] = live ? band_tbl[7 * nbands + threadIdx.x] : 0u;
    }
    const uint32_t term_end = band_tbl[8 * nbands];
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * 32; i += kThreads) {
        const int l = i % 32;
        const uint32_t k = (uint32_t)(row_of_lane(l) + 32 * (i / 32));
        uint32_t at = term_end + (uint32_t)l;               // no band reads the bin: the lane's dump word
        for (uint32_t b = 0; b < 32; ++b)
            if (k >= s_edge[0][b] && k < s_edge[1][b]) at = s_edge[2][b] + (k - s_edge[0][b]);
        term_at[i] = at;
    }
    for (int i = threadIdx.x; i < 32 * 32; i += kThreads) {
        const int q = i / 32, l = i % 32;
        const uint32_t k = (uint32_t)(row_of_lane(l) + 32 * q);     // < 1024
        stw[i] = make_float2(tw[k], tw[kN + k]);
    }
    __syncthreads();

    const int n = lane & 31, s = lane >> 5;                  // phase 1: residue and run; phase 2: row lane and window
    const bool special = n < 2;                              // rows 0 and 16 pair with themselves
    const float inv_norm = 1.0f / (float)(kW / 4);
    float* my_col = tbuf + 32 * s * kRowDw + 2 * n;          // column n of this run's 32 rows
    const float* my_trow = tbuf + lane * kRowDw;             // the row this lane transforms
    const float* my_ctw = ctw + n * kCtwDw;
    float* vbuf = tbuf + s * kPowerDw;                       // power terms of this lane's window, band after band
    const uint32_t* my_term_at = term_at + n;                // where this lane's bins row + 32 q go
    uint32_t b_at = 0, b_full = 0, b_rem = 0;
    float b_div = 1.0f;
    if ((uint32_t)n < nbands) {
        const uint32_t b_lo = band_tbl[n];
        b_at = s_edge[2][n];
        const uint32_t b_hi = band_tbl[nbands + n];
        const uint32_t b_width = b_hi > b_lo ? b_hi - b_lo : 0;
        b_full = b_width >> 3;                               // whole batches of 8 terms
        b_rem = b_width & 7;
        b_div = __uint_as_float(band_tbl[2 * nbands + n]);
    }

    // workgroup b runs on XCD b % 8 (observed; speed only): every XCD owns a contiguous range of runs; a wave
    // claims two neighbouring runs at a time, one per half (the ranges hold an even number of runs)
    const uint32_t xcd = blockIdx.x & 7;
    const uint32_t r_begin = xcd * runs_per_xcd;
    const uint32_t r_end = r_begin + runs_per_xcd < n_runs ? r_begin + runs_per_xcd : n_runs;
    uint32_t* my_ctr = claim_ctr + xcd;
    auto claim = [&]() -> uint32_t {
        uint32_t v = 0;
        if (lane == 0) v = atomicAdd(my_ctr, 2u);
        return r_begin + (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    };

    uint32_t pair = claim();
    while (pair < r_end) {
        const uint32_t next_pair = claim();                              // in flight for a whole run
        const uint32_t run = pair + (uint32_t)s;
        const uint32_t frame = run / kChunksPerFrame, part = run - frame * kChunksPerFrame;
        const uint32_t clip = frame / frames_per_clip;
        const uint32_t fi = frame - clip * frames_per_clip;
        // complex point 0 of the run + this lane's residue
        const int64_t c0 = (int64_t)(((uint64_t)clip * samples_per_clip + (uint64_t)(fi * 128 + part * kChunk) * kStride) >> 1) + n;
        float* out_row = frames + ((uint64_t)frame * 128 + part * kChunk) * nbands + n;

        cplx x[16], P[16], Nw[16];
        load16<FMT, 64, 0>(P, pcm, c0);
        load16<FMT, 64, 0>(x, pcm, c0 + 32);
        st16<1, 0>(P);
        st16<2, 0>(P);
        st16<3, 0>(P);
        st16<4, 0>(P);

        for (int step = 1; step <= kChunk; ++step) {
#pragma unroll
            for (int t = 0; t < 16; ++t) Nw[t] = x[t];
            // the points of the next block go into the registers just vacated: issued now, consumed a whole
            // window later
            if (step < kChunk) load16<FMT, 64, 0>(x, pcm, c0 + 32 * (step + 1));
#ifndef LBAD_EXP_NOPH1
            st16<1, 0>(Nw);
            st16<2, 0>(Nw);
            st16<3, 0>(Nw);
            st16<4, 0>(Nw);
#endif
            // ---- stage 5 and the transpose -----------------------------------------------------------------
            emit_rows<0>(P, Nw, my_col);
            wave_sync();

            // ---- phase 2: this lane's row, bit-reversed column order into the slots ----------------------
            cplx y[32];
#pragma unroll
            for (int i = 0; i < 32; i += 2) {
                const float4 q = *reinterpret_cast<const float4*>(my_trow + 2 * i);
                y[brev5(i)] = mk(q.x, q.y);
                y[brev5(i + 1)] = mk(q.z, q.w);
            }
#ifndef LBAD_EXP_NOCROSS
            {
                int e = 0;
#pragma unroll
                for (int half = 1; half < 32; half <<= 1) {
#pragma unroll
                    for (int jj = 0; jj < half; ++jj) {
                        const f32x2 w = *reinterpret_cast<const f32x2*>(my_ctw + 2 * (e + jj));
#pragma unroll
                        for (int b = 0; b < 32; b += 2 * half) {
                            const cplx u = y[b + jj], v = y[b + jj + half];
                            bfly_w(u, w, v, y[b + jj], y[b + jj + half]);
                        }
                    }
                    e += half;
                }
            }
#endif
            wave_sync();                                                  // every row has been read
            // ---- split pass: bin k = row + 32 q needs Z[N - k] = output 31 - q of the partner row (pair 0:
            //      output 31 - q of row 16 itself, output 32 - q of row 0 itself), for the q some band reads.
            //      Written on (re, im) pairs: packed instructions. -----------------------------------------------
            uint32_t worst = 0;
#pragma unroll
            for (int q = QLO; q < QHI; ++q) {
#ifdef LBAD_EXP_NOSPLIT
                vbuf[my_term_at[32 * q]] = y[q].x + y[31 - q].y;
#else
                {
                    const int j = 31 - q;
                    cplx b;
                    b.x = dpp_pair_swap(y[j].x);
                    b.y = dpp_pair_swap(y[j].y);
                    const cplx own = (lane & 1) ? y[j] : y[j + 1 < 32 ? j + 1 : j];   // (bin 0 of row 0 is never read)
                    if (special) b = own;
                    const cplx a = y[q];
                    const float2 wk = stw[q * 32 + n];
                    cplx sm, df;                                         // a + conj(b) = (sr, si), a - conj(b) = (dr, di)
                    add_conj(a, b, sm, df);
                    // re = fma(wr, di, fma(wi, dr, sr)), im = fma(-wr, dr, fma(wi, di, si))
                    const cplx z = fma2(mk(wk.x, -wk.x), df.yx, fma2(mk(wk.y, wk.y), df, sm));
                    // "if (x > 0) x /= W/4" is min(x * 2^-9, x): one rounding for x > 0, x itself otherwise
                    const cplx zs = z * mk(inv_norm, inv_norm);
                    const cplx zn = mk(fminf(zs.x, z.x), fminf(zs.y, z.y));
                    const cplx sq = zn * zn;
                    const float t = __fadd_rn(sq.x, sq.y);
                    // LBAudioDetective.m:398-401 skips NaN / inf terms: t >= +0.0 unless it is one of them, so as an unsigned
                    // integer every such t lies at or above the bits of +inf -- one v_max_u32 per term keeps watch and the
                    // terms are put right below, in the rare window that has one (was: a class test and a select per term)
                    vbuf[my_term_at[32 * q]] = t;
                    worst = max(worst, __float_as_uint(t));
                }
#endif
            }
            if (__builtin_expect(__any(worst >= 0x7F800000u), 0)) {
                wave_sync();                                              // (a wave's LDS operations execute in order)
                for (int q = QLO; q < QHI; ++q) {
                    float* at = vbuf + my_term_at[32 * q];
                    const float t = *at;
                    *at = (t == t && fabsf(t) != INFINITY) ? t : 0.0f;
                }
            }
            wave_sync();
            // ---- band means in bin order (LBAudioDetective.m:379-405): lane = (band, window) -----------------
            {
                // A band of width w is w / 8 whole batches of 8 terms -- added under a lane mask, no per-term
                // select -- and one partial batch of w % 8 terms read from the lane's own offset.
                const float* vb = vbuf + b_at;
                float v[kMaxTerms], vt[7];
#pragma unroll
                for (uint32_t b = 0; b < kMaxTerms / 8; ++b) {
                    if (b < n_batches) {                                  // wave-uniform
#pragma unroll
                        for (uint32_t q = 0; q < 8; ++q) v[8 * b + q] = vb[8 * b + q];
                    }
                }
#pragma unroll
                for (uint32_t q = 0; q < 7; ++q) vt[q] = vb[8 * b_full + q];
                float p = 0.0f;
#ifdef LBAD_EXP_NOBANDS
                p = v[0] + vt[0];
#else
#pragma unroll
                for (uint32_t b = 0; b < kMaxTerms / 8; ++b) {
                    if (b < b_full) {                                     // per lane
#pragma unroll
                        for (uint32_t q = 0; q < 8; ++q) p = __fadd_rn(p, v[8 * b + q]);
                    }
                }
#pragma unroll
                for (uint32_t q = 0; q < 7; ++q) p = __fadd_rn(p, q < b_rem ? vt[q] : 0.0f);
#endif
                if ((uint32_t)n < nbands) out_row[(uint64_t)(step - 1) * nbands] = __fdiv_rn(p, b_div);
            }
            wave_sync();                                                  // the power terms are consumed
#pragma unroll
            for (int t = 0; t < 16; ++t) P[t] = Nw[t];
        }
        pair = next_pair;
    }
}

}  // namespace

bool rows_stream2_supported(const Plan& p) {
    if (p.window != (uint32_t)kW || p.stride != (uint32_t)kStride || p.bands == 0 || p.bands > 32) return false;
    if (p.table.kmax <= p.table.kmin || p.table.kmin < 1 || p.table.kmax > (uint32_t)kN) return false;
    if (!p.table.ordered) return false;                        // (bands in bin order, no bin in two of them: the terms' layout)
    for (uint32_t b = 0; b < p.bands; ++b)
        if (p.table.hi[b] > p.table.lo[b] && p.table.hi[b] - p.table.lo[b] > (uint32_t)kMaxTerms) return false;
    std::vector<float> re, im;
    make_twiddles(kW, re, im);
    for (int t = 0; t < 32; ++t)
        if (re[(kW / 64) * t] != kTw64Re[t] || im[(kW / 64) * t] != kTw64Im[t]) return false;
    return true;
}

template <int FMT, int QLO, int QHI>
static hipError_t launch_stream2_q(const Plan& plan, const void* d_pcm, uint64_t n_frames, uint64_t samples_per_clip,
                                   uint32_t frames_per_clip, float* d_frames, hipStream_t stream) {
    static PerDevice attr;
    if (attr.changed(kLdsBytes)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rows_stream2_kernel<FMT, QLO, QHI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        if (e != hipSuccess) return e;
    }
    const uint32_t runs_per_xcd = (uint32_t)((n_frames + 7) / 8) * kChunksPerFrame;
    const uint32_t n_runs = (uint32_t)n_frames * kChunksPerFrame;
    uint32_t wg_per_xcd = (uint32_t)device_cu_count() / 8;
    while (wg_per_xcd > 1 && (uint64_t)(wg_per_xcd - 1) * kWaves * 2 >= runs_per_xcd) --wg_per_xcd;
    uint32_t widest = 0;
    for (uint32_t b = 0; b < plan.bands; ++b)
        if (plan.table.hi[b] > plan.table.lo[b] && plan.table.hi[b] - plan.table.lo[b] > widest)
            widest = plan.table.hi[b] - plan.table.lo[b];
    hipError_t e = hipMemsetAsync(plan.d_claim, 0, 8 * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((rows_stream2_kernel<FMT, QLO, QHI>), dim3(wg_per_xcd * 8), dim3(kThreads), kLdsBytes, stream, d_pcm,
                       samples_per_clip, frames_per_clip, n_runs, runs_per_xcd, plan.d_tw, plan.d_bands, plan.bands,
                       plan.table.kmin, plan.table.kmax, (widest + 7) / 8, plan.d_claim, d_frames);
    return hipGetLastError();
}

template <int FMT>
static hipError_t launch_stream2_fmt(const Plan& plan, const void* d_pcm, uint64_t n_frames, uint64_t samples_per_clip,
                                     uint32_t frames_per_clip, float* d_frames, hipStream_t stream) {
    // the default table (5512 Hz: bins 86..758) needs q = 2..23; anything else takes the full range
    if (plan.table.kmin >= 64 && plan.table.kmax <= 768)
        return launch_stream2_q<FMT, 2, 24>(plan, d_pcm, n_frames, samples_per_clip, frames_per_clip, d_frames, stream);
    return launch_stream2_q<FMT, 0, 32>(plan, d_pcm, n_frames, samples_per_clip, frames_per_clip, d_frames, stream);
}

hipError_t launch_rows_stream2(const Plan& plan, const void* d_pcm, uint32_t fmt, uint64_t n_clips, uint64_t samples_per_clip,
                               uint32_t frames_per_clip, float* d_frames, hipStream_t stream) {
    const uint64_t n_frames = n_clips * frames_per_clip;
    if (n_frames == 0) return hipSuccess;
    if (n_frames * kChunksPerFrame > 0x7fffffffull) return hipErrorInvalidValue;
    switch (fmt) {
        case 0: return launch_stream2_fmt<0>(plan, d_pcm, n_frames, samples_per_clip, frames_per_clip, d_frames, stream);
        case 1: return launch_stream2_fmt<1>(plan, d_pcm, n_frames, samples_per_clip, frames_per_clip, d_frames, stream);
        case 2: return launch_stream2_fmt<2>(plan, d_pcm, n_frames, samples_per_clip, frames_per_clip, d_frames, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace lbad
