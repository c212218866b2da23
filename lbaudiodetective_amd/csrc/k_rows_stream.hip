// k_rows_stream.hip -- specialised stage 1 for 4096-sample windows at stride 64 (BASELINE configs[4]:
// 48 kHz, bands reading bins 3..347): windows -> 128 x bands frame rows.
//
// Same arithmetic as k_fft_bands.hip / oracle rfft_exec (radix-2 DIT over the N = 2048 complex points,
// nested-fma butterflies), but a wave WALKS the windows of a frame in time and keeps what consecutive
// windows share.  The hop is 32 complex points, so window i's stage-5 sub-transforms
//     D5(g) = DFT32 { c[g + 64 m] },  g = 32 i + n,  n in [0, 64)
// (c = the clip as complex points) are bit for bit window (i + 1)'s for n >= 32: every D5 is used by two
// windows, as the "even" half of one and the "odd" half of the previous.  Per window the wave therefore
// transforms 32 new residues instead of 64 and never re-reads old points:
//
//   phase 1  lane (n, h), n = lane & 31, h = lane >> 5: 16 points c[g + 64 h + 128 m'] from L2 -> D4 (DIT
//            stages 1..4 in registers, compile-time twiddles) -> the halves trade half of their D4s (one
//            v_permlane32_swap per float, no copies: lane (n, 0) ends up with both inputs of the stage-5
//            butterflies 0..7, lane (n, 1) with those of 8..15) and stage 5 leaves D5(g)[k], k in
//            {8 h + j, 16 + 8 h + j}, in lane (n, h) -> stage 6 with the D5 of the previous step,
//            E(g)[k] = P[k] +- W_64^k Nw[k], written to the wave's LDS transpose (row k, column n).
//   phase 2  lane = one row k64: the 32-point cross transform over n (DIT stages 7..11, twiddles from a
//            per-lane LDS table), pruned to the 12 outputs q in {0..5, 26..31} that bins < 384 and their
//            mirror bins need; partner rows (a, 64 - a) sit in neighbouring lanes and trade their six
//            high outputs by DPP, which makes the split pass, the positive-only normalisation and the
//            power term lane-local.  Power terms -> LDS, band sums in bin order as in the reference.
//
// A workgroup is 8 independent waves (two per SIMD) that share only the twiddle tables; there is no
// workgroup barrier after start-up.  Frames are claimed from per-XCD counters, so the overlapping
// spans of neighbouring frames meet in one L2.
#include "stream_common.hpp"

namespace lbad {
namespace {

using namespace lane64;
using namespace stream;

constexpr int kW = 4096;
constexpr int kN = kW / 2;
constexpr int kStride = 64;
constexpr int kWaves = 8;
#ifndef LBAD_EXP_CHUNK
#define LBAD_EXP_CHUNK 16
#endif
constexpr int kChunk = LBAD_EXP_CHUNK;                    // windows a wave walks before it claims again (see the kernel)
constexpr int kChunksPerFrame = 128 / kChunk;
constexpr int kThreads = kWaves * 64;
constexpr int kRowDw = 68;                    // 32 complex + 16 B: lanes 0..15 of a b128 read hit disjoint banks
constexpr int kTDw = 64 * kRowDw;             // transpose of one window (one wave)
constexpr int kCrossTw = 27;                  // cross-stage twiddles a lane uses: 1 + 2 + 4 + 8 + 12
constexpr int kCtwDw = 66;                    // their pitch: the 32 lanes of a ds_read_b64 on 64 different banks (at 68 lanes l, l + 16 shared theirs)
constexpr int kQ = 6;                         // low outputs per row: bins a + 64 q < 384
constexpr int kMaxBin = 64 * kQ;
constexpr int kMaxTerms = 48;                 // bins of the widest band (the band sums are unrolled this far)
// power terms of a window: band after band, every band starting on a bank of its own (BandTable::term_at -- see
// k_rows_stream2.hip; by bin number until round 5: 34 % of the LDS-active cycles were bank conflicts), a dump word per lane
// and the read overrun of the last band behind them
constexpr int kPowerDw = kTDw / 2;
static_assert(kMaxBin + 32 * 31 + 64 + kMaxTerms + 7 <= kPowerDw, "skewed power terms of a window fit half the transpose area");
constexpr int kP1 = 24;                       // phase-1 twiddles per half: 8 of stage 5, 16 of stage 6
constexpr int kLdsDw = kWaves * kTDw + 64 * kCtwDw + kQ * 64 * 2 + 2 * kP1 * 2;
constexpr int kLdsBytes = kLdsDw * 4;         // 159 744 B: one workgroup per CU
static_assert(kLdsBytes <= 160 * 1024, "LDS budget");
static_assert(2 * kPowerDw <= kTDw, "the power terms of two windows reuse the transpose area");

// row of lane l: lanes 2 p, 2 p + 1 hold the partner rows (p, 64 - p); pair 0 is (0, 32)
__device__ __forceinline__ int row_of_lane(int l) {
    const int p = l >> 1;
    return (l & 1) == 0 ? p : (p == 0 ? 32 : 64 - p);
}

template <int FMT>
__global__ __launch_bounds__(kThreads, 2) void rows_stream_kernel(const void* __restrict__ pcm, uint64_t samples_per_clip,
                                                                  uint32_t frames_per_clip, uint32_t n_frames,
                                                                  uint32_t frames_per_xcd, const float* __restrict__ tw,
                                                                  const uint32_t* __restrict__ band_tbl, uint32_t nbands,
                                                                  uint32_t kmin, uint32_t kmax, uint32_t n_batches,
                                                                  uint32_t* __restrict__ claim_ctr,
                                                                  float* __restrict__ frames) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    float* tbuf = smem + wave * kTDw;                       // this wave's transpose / power terms
    float* ctw = smem + kWaves * kTDw;                      // [lane][27 complex], row pitch kCtwDw
    float2* stw = reinterpret_cast<float2*>(ctw + 64 * kCtwDw);   // [q][lane]
    float2* p1tw = stw + kQ * 64;                           // [h][24]: stage 5 W_32^(8 h + j), stage 6 W_64^k(h, kk)

    // ---- once per workgroup: tables --------------------------------------------------------------------
    for (int i = threadIdx.x; i < 64 * kCrossTw; i += kThreads) {
        const int l = i / kCrossTw, e = i % kCrossTw;
        const int a = row_of_lane(l);
        int s, jj;                                          // cross stage s = 1..5 (overall stage 6 + s), butterfly jj
        if (e < 1) { s = 1; jj = 0; }
        else if (e < 3) { s = 2; jj = e - 1; }
        else if (e < 7) { s = 3; jj = e - 3; }
        else if (e < 15) { s = 4; jj = e - 7; }
        else { s = 5; jj = e - 15 < 6 ? e - 15 : e - 15 + 4; }     // jj = 0..5, 10..15
        const uint32_t ti = (uint32_t)(a + 64 * jj) << (6 - s);     // W_(64 * 2^s)^(a + 64 jj)
        ctw[l * kCtwDw + 2 * e] = tw[ti];
        ctw[l * kCtwDw + 2 * e + 1] = tw[kN + ti];
    }
    for (int i = threadIdx.x; i < kQ * 64; i += kThreads) {
        const int q = i / 64, l = i % 64;
        const uint32_t k = (uint32_t)(row_of_lane(l) + 64 * q);     // < 2048
        stw[i] = make_float2(tw[k], tw[kN + k]);
    }
    for (int i = threadIdx.x; i < 2 * kP1; i += kThreads) {
        const int h = i / kP1, e = i % kP1;
        uint32_t ti;
        if (e < 8) {
            ti = (uint32_t)(8 * h + e) * (kW / 32);         // stage 5: W_32^(8 h + j)
        } else {
            const int kk = e - 8;
            ti = (uint32_t)(8 * h + kk + (kk >= 8 ? 8 : 0)) * (kW / 64);   // stage 6: W_64^k, k = k(h, kk)
        }
        p1tw[i] = make_float2(tw[ti], tw[kN + ti]);
    }
    __syncthreads();

    const int n = lane & 31, h = lane >> 5;
    const int my_row = row_of_lane(lane);
    const bool special = lane < 2;                           // rows 0 and 32 pair with themselves
    const float inv_norm = 1.0f / (float)(kW / 4);
    const float2* my_p1 = p1tw + kP1 * h;
    float* my_col = tbuf + n * 2;                            // column n of the transpose (row pitch kRowDw)
    const float* my_trow = tbuf + lane * kRowDw;             // the row this lane transforms (stored by destination lane)
    const float* my_ctw = ctw + lane * kCtwDw;
    // Lane (n, h) holds D5[k] for k = k(h, kk) = 8 h + kk (kk < 8), 8 + 8 h + kk (kk >= 8) and emits rows k and
    // k + 32.  A row is stored at the slot of the lane that transforms it (the inverse of row_of_lane: 0 -> 0,
    // 32 -> 1, k < 32 -> 2 k, else 2 (64 - k) + 1), which is affine in kk within each group of eight.
    // (bases chosen so that every store is base + a non-negative compile-time offset)
    float* col_p0 = my_col + 16 * h * kRowDw;                //  k      -> lane 16 h + 2 kk               (kk < 8)
    float* col_p1 = my_col + (16 + 16 * h) * kRowDw;         //  k      -> lane 16 + 16 h + 2 kk          (kk >= 8)
    float* col_m0 = my_col + ((h ? 49 : 65) - 14) * kRowDw;  //  k + 32 -> lane (h ? 49 : 65) - 2 kk      (kk < 8), + (14 - 2 kk)
    float* col_m1 = my_col + ((h ? 33 : 49) - 30) * kRowDw;  //  k + 32 -> lane (h ? 33 : 49) - 2 kk      (kk >= 8), + (30 - 2 kk)
    float* col_m00 = my_col + (h ? 49 : 1) * kRowDw;         //  row 32 (h = 0, kk = 0) -> lane 1
    // where the power terms of this lane's six low bins go inside a window's area: the word the band that reads the bin keeps
    // for it, or the lane's dump word (fixed for the life of the workgroup: six registers)
    const uint32_t term_end = band_tbl[8 * nbands];
    uint32_t at[kQ];
#pragma unroll
    for (int q = 0; q < kQ; ++q) {
        const uint32_t k = (uint32_t)(my_row + 64 * q);
        at[q] = term_end + (uint32_t)lane;
        if (k != 0)
            for (uint32_t b = 0; b < nbands; ++b) {
                const uint32_t lo = band_tbl[b], hi = band_tbl[nbands + b];
                if (k >= lo && k < hi) at[q] = band_tbl[7 * nbands + b] + (k - lo);
            }
    }
    float* vbuf = tbuf;                                      // power terms: [2 windows][kPowerDw]
    uint32_t b_at = 0, b_full = 0, b_rem = 0;                 // band lane & 31 (both halves: two windows per pass)
    float b_div = 1.0f;
    if ((uint32_t)(lane & 31) < nbands) {
        const uint32_t b_lo = band_tbl[lane & 31];
        b_at = band_tbl[7 * nbands + (lane & 31)];
        const uint32_t b_hi = band_tbl[nbands + (lane & 31)];
        const uint32_t b_width = b_hi > b_lo ? b_hi - b_lo : 0;
        b_full = b_width >> 3;                                // whole batches of 8 terms
        b_rem = b_width & 7;
        b_div = __uint_as_float(band_tbl[2 * nbands + (lane & 31)]);
    }

    // workgroup b runs on XCD b % 8 (observed; speed only): every XCD owns a contiguous range of frames
    const uint32_t xcd = blockIdx.x & 7;
    const uint32_t f_begin = xcd * frames_per_xcd;
    const uint32_t f_end = f_begin + frames_per_xcd < n_frames ? f_begin + frames_per_xcd : n_frames;
    uint32_t* my_ctr = claim_ctr + xcd;
    auto claim = [&]() -> uint32_t {
        uint32_t v = 0;
        if (lane == 0) v = atomicAdd(my_ctr, 1u);
        return f_begin + (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    };

    // D5 of block `blk` of the frame that starts at complex point `c0`: this lane's half
    auto d5_block = [&](const cplx (&x_in)[16], cplx (&out)[16], const float2 (&wt)[kP1]) {
        cplx x[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) x[t] = x_in[t];
#ifndef LBAD_EXP_NOFFT16
        st16<1, 0>(x);
        st16<2, 0>(x);
        st16<3, 0>(x);
        st16<4, 0>(x);
#endif
        // The halves trade D4s: after swapping x[j] (upper half) with x[8 + j] (lower half) every lane holds in
        // (x[j], x[8 + j]) = (u, v) = (D4(g), D4(g + 64)) at index 8 h + j, the inputs of stage-5 butterfly 8 h + j.
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#ifndef LBAD_EXP_NOSWAP
            const auto rx = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[j].x), __float_as_uint(x[8 + j].x), false, false);
            const auto ry = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[j].y), __float_as_uint(x[8 + j].y), false, false);
            const cplx u = mk(__uint_as_float(rx[0]), __uint_as_float(ry[0]));
            const cplx v = mk(__uint_as_float(rx[1]), __uint_as_float(ry[1]));
#else
            const cplx u = x[j], v = x[8 + j];
#endif
            const float2 w = wt[j];                                      // W_32^(8 h + j)
            bfly_w(u, mk(w.x, w.y), v, out[j], out[8 + j]);              // D5[8 h + j], D5[8 h + j + 16]
        }
    };

    // The unit of work is a run of kChunk consecutive windows, claimed in order: the waves of an XCD are at any
    // time inside a few dozen neighbouring frames, so the 16 KB sliding windows they read overlap and stay in
    // the XCD's L2 (whole frames per wave put 4 MB of distinct windows against a 4 MB L2: every load missed).
    // Price: one extra D5 block per run.
    uint32_t chunk = claim();
    while (chunk < f_end) {
        const uint32_t next_chunk = claim();                             // in flight for a whole run
        const uint32_t frame = chunk / kChunksPerFrame, part = chunk - frame * kChunksPerFrame;
        const uint32_t clip = frame / frames_per_clip;
        const uint32_t fi = frame - clip * frames_per_clip;
        // complex point 0 of the run; lane (n, h) reads points g + 64 h + 128 m'
        const int64_t c0 = (int64_t)(((uint64_t)clip * samples_per_clip + (uint64_t)(fi * 128 + part * kChunk) * kStride) >> 1) + n + 64 * h;
        float* out_row = frames + ((uint64_t)frame * 128 + part * kChunk) * nbands + n;

        cplx xa[16], xb[16], P[16], Nw[16];
        load16<FMT, 128, 0>(xa, pcm, c0);
        load16<FMT, 128, 0>(xb, pcm, c0 + 32);
        {
            float2 wt0[kP1];
#pragma unroll
            for (int i = 0; i < 8; ++i) wt0[i] = my_p1[i];
            d5_block(xa, P, wt0);
        }

        // one window: x holds the points of block `step`, xn receives those of block `step + 1`
        auto window_step = [&](int step, cplx (&x)[16], cplx (&xn)[16], cplx (&Pp)[16], cplx (&Nn)[16], float (&pw_out)[kQ]) {
            // the points of the next block: issued now, consumed a whole step later (the scheduler must not
            // sink them towards their use to save registers: that would expose the memory latency)
            if (step < kChunk) load16<FMT, 128, 0>(xn, pcm, c0 + 32 * (step + 1));
            // this half's stage-5 / stage-6 twiddles: all reads in flight before the arithmetic starts (left to
            // itself the scheduler fetches each one right before its butterfly and waits for it, 24 times)
            float2 wt[kP1];
#pragma unroll
            for (int i = 0; i < kP1; ++i) wt[i] = my_p1[i];
            __builtin_amdgcn_sched_barrier(0);
            d5_block(x, Nn, wt);
            // ---- stage 6 and the transpose: rows k(h, kk) and k + 32, column n ------------------------------
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const float2 w = wt[8 + kk];                             // W_64^k
                cplx ep, em;
                bfly_w(Pp[kk], mk(w.x, w.y), Nn[kk], ep, em);
                *(lds_vf32x2*)((kk < 8 ? col_p0 : col_p1) + 2 * kk * kRowDw) = ep;
                *(lds_vf32x2*)(kk == 0 ? col_m00 : (kk < 8 ? col_m0 + (14 - 2 * kk) * kRowDw : col_m1 + (30 - 2 * kk) * kRowDw)) = em;
            }
            wave_sync();

#ifdef LBAD_EXP_NOPHASE2
#pragma unroll
            for (int q = 0; q < kQ; ++q) pw_out[q] = my_trow[q];
            return;
#endif
            // ---- phase 2: this lane's row, bit-reversed column order into the slots ----------------------
            cplx y[32];
#pragma unroll
            for (int i = 0; i < 32; i += 2) {
                const float4 q = *reinterpret_cast<const float4*>(my_trow + 2 * i);
                y[brev5(i)] = mk(q.x, q.y);
                y[brev5(i + 1)] = mk(q.z, q.w);
            }
            // cross stages 1..3 in full
            {
                const f32x2 w0 = *reinterpret_cast<const f32x2*>(my_ctw);
#pragma unroll
                for (int b = 0; b < 32; b += 2) {
                    const cplx u = y[b], v = y[b + 1];
                    bfly_w(u, w0, v, y[b], y[b + 1]);
                }
            }
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const f32x2 w = *reinterpret_cast<const f32x2*>(my_ctw + 2 * (1 + jj));
#pragma unroll
                for (int b = 0; b < 32; b += 4) {
                    const cplx u = y[b + jj], v = y[b + jj + 2];
                    bfly_w(u, w, v, y[b + jj], y[b + jj + 2]);
                }
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const f32x2 w = *reinterpret_cast<const f32x2*>(my_ctw + 2 * (3 + jj));
#pragma unroll
                for (int b = 0; b < 32; b += 8) {
                    const cplx u = y[b + jj], v = y[b + jj + 4];
                    bfly_w(u, w, v, y[b + jj], y[b + jj + 4]);
                }
            }
            // cross stage 4, pruned: per 16-block the outputs p in {0..5} ("+") and {10..15} ("-")
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const f32x2 w = *reinterpret_cast<const f32x2*>(my_ctw + 2 * (7 + jj));
#pragma unroll
                for (int b = 0; b < 32; b += 16) {
                    const cplx u = y[b + jj], v = y[b + jj + 8];
                    if (jj >= 2 && jj < 6) bfly_w(u, w, v, y[b + jj], y[b + jj + 8]);
                    else if (jj < 6) y[b + jj] = madd_w(u, w, v);
                    else y[b + jj + 8] = msub_w(u, w, v);
                }
            }
            // cross stage 5, pruned: outputs q in {0..5} ("+") and {26..31} ("-" of pairs 10..15).  The high
            // outputs first: bin k = row + 64 q needs Z[N - k] = output 31 - q of the PARTNER row, which sits in
            // the neighbouring lane (pair 0: output 31 - q of row 32 itself, output 32 - q of row 0 itself).
            cplx bsel[kQ];                                               // bsel[q] = Z[N - (row + 64 q)]
            {
                cplx hi[kQ];
#pragma unroll
                for (int j = 0; j < kQ; ++j) {
                    const f32x2 wb = *reinterpret_cast<const f32x2*>(my_ctw + 2 * (21 + j));
                    hi[j] = msub_w(y[10 + j], wb, y[26 + j]);            // output 26 + j
                }
#pragma unroll
                for (int q = 0; q < kQ; ++q) {
                    const int j = kQ - 1 - q;                            // output 31 - q
                    cplx b;
                    b.x = dpp_pair_swap(hi[j].x);
                    b.y = dpp_pair_swap(hi[j].y);
                    const cplx own = (lane & 1) ? hi[j] : hi[j + 1 < kQ ? j + 1 : j];   // (bin 0 of row 0 is never read)
                    bsel[q] = special ? own : b;
                }
            }
            // ---- the low outputs one at a time, each straight into the split pass.  Written on (re, im) pairs
            //      so that it compiles to packed operations; each half is the oracle's fmaf / mul / add. ------
#pragma unroll
            for (int q = 0; q < kQ; ++q) {
                const f32x2 wa = *reinterpret_cast<const f32x2*>(my_ctw + 2 * (15 + q));
                const cplx a = madd_w(y[q], wa, y[q + 16]);              // output q
                const cplx b = bsel[q];
                const float2 wk = stw[q * 64 + lane];
                cplx sm, df;                                             // a + conj(b) = (sr, si), a - conj(b) = (dr, di)
                add_conj(a, b, sm, df);
                // re = fma(wr, di, fma(wi, dr, sr)), im = fma(-wr, dr, fma(wi, di, si))
                const cplx z = fma2(mk(wk.x, -wk.x), df.yx, fma2(mk(wk.y, wk.y), df, sm));
                // "if (x > 0) x /= W/4" is min(x * 2^-10, x): one rounding for x > 0, x itself otherwise
                const cplx zs = z * mk(inv_norm, inv_norm);
                const cplx zn = mk(fminf(zs.x, z.x), fminf(zs.y, z.y));
                const cplx sq = zn * zn;
                const float t = __fadd_rn(sq.x, sq.y);
                pw_out[q] = (t == t && fabsf(t) != INFINITY) ? t : 0.0f;  // LBAudioDetective.m:398-401, at the source
            }
        };

        // Band means (LBAudioDetective.m:379-405) of TWO windows at a time: lanes 0..31 sum the bins of window
        // w0, lanes 32..63 those of w0 + 1, lane & 31 = band.  The sum is sequential in bin order by
        // definition; one pass of (load, select, add) instructions serves both windows.
        auto band_sums = [&](const float (&pw0)[kQ], const float (&pw1)[kQ], int w0) {
            wave_sync();                                                  // every row of the last window has been read
#pragma unroll
            for (int q = 0; q < kQ; ++q) {
                vbuf[at[q]] = pw0[q];
                vbuf[kPowerDw + at[q]] = pw1[q];
            }
            wave_sync();
            const float* vb = vbuf + h * kPowerDw + b_at;
            // A band of width w is w / 8 whole batches of 8 terms -- added under a lane mask, no per-term select --
            // and one partial batch of w % 8 terms read from the lane's own offset.
            float v[kMaxTerms], vt[7];
#pragma unroll
            for (uint32_t b = 0; b < kMaxTerms / 8; ++b) {
                if (b < n_batches) {                                      // wave-uniform
#pragma unroll
                    for (uint32_t q = 0; q < 8; ++q) v[8 * b + q] = vb[8 * b + q];
                }
            }
#pragma unroll
            for (uint32_t q = 0; q < 7; ++q) vt[q] = vb[8 * b_full + q];
            float p = 0.0f;
#pragma unroll
            for (uint32_t b = 0; b < kMaxTerms / 8; ++b) {
                if (b < b_full) {                                         // per lane
#pragma unroll
                    for (uint32_t q = 0; q < 8; ++q) p = __fadd_rn(p, v[8 * b + q]);
                }
            }
#pragma unroll
            for (uint32_t q = 0; q < 7; ++q) p = __fadd_rn(p, q < b_rem ? vt[q] : 0.0f);
            if ((uint32_t)n < nbands) out_row[(uint64_t)(w0 + h) * nbands] = __fdiv_rn(p, b_div);
            wave_sync();                                                  // the power terms are consumed
        };

        float pwa[kQ], pwb[kQ];
        // one window per iteration (not unrolled: two steps interleaved by the scheduler need more registers
        // than a wave has); the hand-over of the 64 + 6 values costs 70 moves per window
        for (int step = 1; step <= kChunk; ++step) {
            window_step(step, xb, xa, P, Nw, pwb);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                xb[t] = xa[t];
                P[t] = Nw[t];
            }
            if ((step & 1) == 0) {
                band_sums(pwa, pwb, step - 2);
            } else {
#pragma unroll
                for (int q = 0; q < kQ; ++q) pwa[q] = pwb[q];
            }
        }
        chunk = next_chunk;
    }
}

}  // namespace

bool rows_stream_supported(const Plan& p) {
    if (p.window != (uint32_t)kW || p.stride != (uint32_t)kStride || p.bands == 0 || p.bands > 32) return false;
    if (p.table.kmax <= p.table.kmin || p.table.kmin < 1 || p.table.kmax > (uint32_t)kMaxBin) return false;
    if (!p.table.ordered) return false;                        // (bands in bin order, no bin in two of them: the terms' layout)
    for (uint32_t b = 0; b < p.bands; ++b)
        if (p.table.hi[b] > p.table.lo[b] && p.table.hi[b] - p.table.lo[b] > (uint32_t)kMaxTerms) return false;
    std::vector<float> re, im;
    make_twiddles(kW, re, im);
    for (int t = 0; t < 32; ++t)
        if (re[(kW / 64) * t] != kTw64Re[t] || im[(kW / 64) * t] != kTw64Im[t]) return false;
    return true;
}

template <int FMT>
static hipError_t launch_stream_fmt(const Plan& plan, const void* d_pcm, uint64_t n_frames_in, uint64_t samples_per_clip,
                                    uint32_t frames_per_clip, float* d_frames, hipStream_t stream) {
    uint64_t n_frames = n_frames_in;
    static PerDevice attr;
    if (attr.changed(kLdsBytes)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rows_stream_kernel<FMT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        if (e != hipSuccess) return e;
    }
    const uint32_t frames_per_xcd = (uint32_t)((n_frames + 7) / 8) * kChunksPerFrame;     // in runs of kChunk windows
    n_frames *= kChunksPerFrame;
    const uint32_t waves_per_xcd = (uint32_t)device_cu_count() / 8 * kWaves;
    // no more workgroups than an XCD has frames to hand out (a wave that finds no frame exits at once)
    uint32_t wg_per_xcd = (uint32_t)device_cu_count() / 8;
    while (wg_per_xcd > 1 && (uint64_t)(wg_per_xcd - 1) * kWaves >= frames_per_xcd) --wg_per_xcd;
    (void)waves_per_xcd;
    uint32_t widest = 0;
    for (uint32_t b = 0; b < plan.bands; ++b)
        if (plan.table.hi[b] > plan.table.lo[b] && plan.table.hi[b] - plan.table.lo[b] > widest)
            widest = plan.table.hi[b] - plan.table.lo[b];
    hipError_t e = hipMemsetAsync(plan.d_claim, 0, 8 * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(rows_stream_kernel<FMT>, dim3(wg_per_xcd * 8), dim3(kThreads), kLdsBytes, stream, d_pcm,
                       samples_per_clip, frames_per_clip, (uint32_t)n_frames, frames_per_xcd, plan.d_tw, plan.d_bands,
                       plan.bands, plan.table.kmin, plan.table.kmax, (widest + 7) / 8, plan.d_claim, d_frames);
    return hipGetLastError();
}

hipError_t launch_rows_stream(const Plan& plan, const void* d_pcm, uint32_t fmt, uint64_t n_clips, uint64_t samples_per_clip,
                              uint32_t frames_per_clip, float* d_frames, hipStream_t stream) {
    const uint64_t n_frames = n_clips * frames_per_clip;
    if (n_frames == 0) return hipSuccess;
    if (n_frames * kChunksPerFrame > 0x7fffffffull) return hipErrorInvalidValue;
    switch (fmt) {
        case 0: return launch_stream_fmt<0>(plan, d_pcm, n_frames, samples_per_clip, frames_per_clip, d_frames, stream);
        case 1: return launch_stream_fmt<1>(plan, d_pcm, n_frames, samples_per_clip, frames_per_clip, d_frames, stream);
        case 2: return launch_stream_fmt<2>(plan, d_pcm, n_frames, samples_per_clip, frames_per_clip, d_frames, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace lbad
