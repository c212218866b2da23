// k_rows_full.hip -- specialised stage 1 for 256- to 2048-sample windows with ANY band table and ANY stride
// (the reference's default 5512 Hz / 2048 among them): windows -> 128 x bands frame rows.
//
// Same arithmetic as k_fft_bands.hip / oracle rfft_exec (radix-2 DIT, nested-fma butterflies), laid out
// like k_rows_pruned.hip but without pruning:
//
//   * L = N / 64 lanes own one window (N = W / 2 complex points; L = 8 or 16).  Lane r holds the 64
//     points z[r + L m] and runs DIT stages 1..6 in registers with compile-time twiddles.
//   * The remaining log2 L stages are, for every k64, an L-point transform over the lanes' values
//     X_r[k64].  They cross lanes through a per-wave LDS transpose, one row per lane and pass: in pass
//     p lane j receives the L values of "its" row, runs the L-point DIT in registers with the row's
//     L - 1 twiddles (an LDS table built once per workgroup) and ends up with the bins k64 + 64 u.
//   * Lane j is given rows in pairs (a, 64 - a) -- and (0, 32) -- because bin k of the real transform
//     needs Z[k] and Z[N - k]: with that pairing the split pass, the positive-only normalisation and
//     the power term are lane-local.  Power terms of the bins the bands read go to LDS, band sums run
//     in bin order as in the reference.
//   * A workgroup (4 waves) owns 256 / L consecutive windows of one frame and reads their PCM span from
//     HBM once.
#include "internal.hpp"
#include "fft64_lane.hpp"

#include <type_traits>

namespace lbad {
namespace {

using namespace lane64;

constexpr int kStride = 64;
constexpr int kWaves = 4;
constexpr int kThreads = kWaves * 64;

template <int LOG2L> struct Shape {
    static constexpr int L = 1 << LOG2L;               // lanes per window
    static constexpr int N = 64 * L;                   // complex points
    static constexpr int W = 2 * N;                    // samples per window
    static constexpr int LOG2W = 7 + LOG2L;
    static constexpr int R = 64 / L;                   // rows (k64 values) per lane = transpose passes
    static constexpr int WPW = 64 / L;                 // windows per wave
    static constexpr int UW = kWaves * WPW;            // windows per workgroup
    // Stride 64 (S64): the L lanes of a window read 2 L consecutive floats; the windows of a 32-lane group must land
    // on disjoint banks: skew every run of 64 samples by 2 L dwords.  Any other stride (run-time value): the span
    // is stored as it is -- windows that overlap read the SAME addresses (broadcast, no conflict).
    static constexpr int kSkew = 2 * L;
    static constexpr int kSpan = (UW - 1) * kStride + W;
    static constexpr int kSpanDw = kSpan + kSkew * (kSpan >> 6);
    static constexpr uint32_t span_len(uint32_t stride) { return (uint32_t)(UW - 1) * stride + (uint32_t)W; }
    static constexpr uint32_t span_dw(uint32_t stride, bool s64) {
        return s64 ? (uint32_t)kSpanDw : ((span_len(stride) + 63u) & ~63u);
    }
    static constexpr int kRowDw = 2 * L + 4;           // transpose row: L values of 8 B, padded by 16 B
    // one pass of one window.  L = 16: a ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31},
    // ... (MI355X_MICROARCH.md, LDS) -- half of one window's rows and half of the next one's; with the windows a whole
    // number of 256-byte bank rows apart the two halves fall on different 16-byte slots (rows r and r + 8 of ONE window
    // do: 9 r mod 16).  Until round 5 the windows lay 32 words further apart and every such read met a 2-way conflict.
    static constexpr int kWinDw = L == 16 ? L * kRowDw : L * kRowDw + (L == 8 ? 16 : 32);
    static_assert(L != 16 || (L * kRowDw) % 64 == 0, "windows a whole number of bank rows apart");
    // cross-lane twiddles of one row: (L - 1) x 8 B; the 16 rows 2 r + Q a half-wave reads together lie 4 banks apart either
    // way (L = 16: 60 r mod 64 without padding, 68 r with the two words the smaller shapes keep)
    static constexpr int kTwRowDw = L == 16 ? 2 * L - 2 : 2 * L + 2;
};

template <class F, int... I>
__device__ __forceinline__ void static_for_seq(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int COUNT, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_seq(f, std::make_integer_sequence<int, COUNT>{}); }

// The split-pass terms no band of the DEFAULT table (5512 Hz, 2048-sample windows, 32 bands: bins 86..758) reads from ANY
// lane, as bits (q * 16 + u) * 2 + half (BandTable::unread_terms16): 14 of the 64 places -- the lanes' bins 0..31 and
// 993..1024 and most of 768..992; lane 0's slot 0 pairs rows 0 and 32 with themselves and keeps six more alive.
// rows_full_kernel<4, ...> has an instance that does not compute them (- 7 % vector instructions); any table whose
// unread set contains this one may use it.
constexpr uint64_t kDefaultUnread16 = 0xd50000ab000000a9ull;

// words between the power terms of two windows of a wave: the terms, a dump word per lane of the window (the six words a
// band's last partial batch reads past its end lie in them too), odd (see the kernel)
__host__ __device__ constexpr uint32_t term_pitch(uint32_t term_end) { return (term_end + 16u) | 1u; }
// words of a wave's LDS area: the transpose of a pass, later the power terms of its windows (16-byte granules: the transpose
// is read as float4)
template <int LOG2L> __host__ __device__ constexpr uint32_t wave_words(uint32_t term_end) {
    using S = Shape<LOG2L>;
    const uint32_t a = (uint32_t)(S::WPW * S::kWinDw), b = (uint32_t)S::WPW * term_pitch(term_end);
    return ((a > b ? a : b) + 3u) & ~3u;
}

// L-point DIT over the lanes' values of one row.  v[] is in bit-reversed lane order on entry (slot i
// holds lane brev(i)) and in natural order of u on exit: v[u] = X[k64 + 64 u].  tw[] holds the row's
// twiddles stage by stage: [1][2][4]...; butterfly jj of a stage uses W^(k64 + 64 jj).
template <int L>
__device__ __forceinline__ void cross_fft(cplx (&v)[L], const float* tw) {
    int t0 = 0;
#pragma unroll
    for (int half = 1; half < L; half <<= 1) {
        cplx w[L / 2];
#pragma unroll
        for (int jj = 0; jj < half; ++jj) w[jj] = *reinterpret_cast<const f32x2*>(tw + 2 * (t0 + jj));
#pragma unroll
        for (int b = 0; b < L; b += 2 * half)
#pragma unroll
            for (int jj = 0; jj < half; ++jj) {
                // a + w c and a - w c: fma(wr, c, fma(-+wi, c.yx, a)) per half (fft64_lane.hpp: the twiddle pair as it is read)
                const cplx a = v[b + jj], c = v[b + jj + half];
                bfly_w(a, w[jj], c, v[b + jj], v[b + jj + half]);
            }
        t0 += half;
    }
}

template <int L> __device__ __forceinline__ constexpr int brevL(int v) {
    int r = 0;
    for (int b = 1; b < L; b <<= 1) {
        r = (r << 1) | (v & 1);
        v >>= 1;
    }
    return r;
}

template <int LOG2L, bool S64, int M>
__device__ __forceinline__ void load_points64(cplx (&x)[64], const float* src) {
    using S = Shape<LOG2L>;
    if constexpr (M < 64) {
        // register slot M holds point m = brev6(M) of this lane: samples 2 (r + L m), 2 (r + L m) + 1 of
        // the window, i.e. 2 L m floats after the lane base plus the skew of the 64-sample runs crossed
        // (2 r + (2 L m mod 64) < 64, so the lane offset never crosses a run itself)
        constexpr int m = brev6(M);
        constexpr int off = 2 * S::L * m + (S64 ? S::kSkew * ((2 * S::L * m) >> 6) : 0);
        x[M] = *(const lds_vf32x2*)(src + off);
        load_points64<LOG2L, S64, M + 1>(x, src);
    }
}

// pass P of the transpose: destination lane j receives row  rho(j, P) = P even ? j R/2 + P/2
//                                                                     : 64 - (j R/2 + P/2)  (32 for slot 0)
template <int LOG2L, int P, int J>
__device__ __forceinline__ void store_pass(const cplx (&x)[64], float* tcol) {
    using S = Shape<LOG2L>;
    if constexpr (J < S::L) {
        constexpr int slot = J * (S::R / 2) + P / 2;
        constexpr int row = (P & 1) == 0 ? slot : (slot == 0 ? 32 : 64 - slot);
        *(lds_vf32x2*)(tcol + J * S::kRowDw) = x[row];
        store_pass<LOG2L, P, J + 1>(x, tcol);
    }
}

typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

// span of one unit -> LDS, every run of 64 samples followed by kSkew pad dwords.  float32: one
// global_load_lds_dword per run (no registers, the wave does not wait); int16 / int32 convert in registers.
template <int LOG2L, int FMT, bool S64>
__device__ __forceinline__ void span_to_lds(const void* __restrict__ pcm_raw, uint64_t first, float* span, int wave, int lane,
                                            uint32_t stride) {
    using S = Shape<LOG2L>;
    if constexpr (!S64) {
        // any stride, float32: runs of 64 samples, unskewed; the last run may be partial (never read past the span:
        // the clip buffer may end there)
        static_assert(FMT == 0, "strides other than 64 take float32 input");
        const uint32_t len = S::span_len(stride);
        const float* src = static_cast<const float*>(pcm_raw) + first;
        for (uint32_t run = wave; 64u * run < len; run += kWaves) {
            if (64u * run + (uint32_t)lane < len)
                __builtin_amdgcn_global_load_lds((gvoid_t*)(src + 64u * run + lane), (lvoid_t*)(span + 64u * run), 4, 0, 0);
        }
        return;
    }
    constexpr int kPitch = 64 + S::kSkew;
    if constexpr (FMT == 0) {
        const float* src = static_cast<const float*>(pcm_raw) + first + lane + 64 * wave;
        float* dst = span + kPitch * wave;
        constexpr int kRuns = S::kSpan / 64, kFull = kRuns / kWaves;
#pragma unroll
        for (int i = 0; i < kFull; ++i)
            __builtin_amdgcn_global_load_lds((gvoid_t*)(src + 64 * kWaves * i), (lvoid_t*)(dst + kPitch * kWaves * i), 4, 0, 0);
        if (wave < kRuns - kFull * kWaves)
            __builtin_amdgcn_global_load_lds((gvoid_t*)(src + 64 * kWaves * kFull), (lvoid_t*)(dst + kPitch * kWaves * kFull), 4, 0, 0);
    } else if constexpr (FMT == 1) {
        const int16_t* src = static_cast<const int16_t*>(pcm_raw) + first;
        for (int s = threadIdx.x; s < S::kSpan; s += kThreads) span[s + S::kSkew * (s >> 6)] = (float)src[s] * (1.0f / 32768.0f);
    } else {
        const int32_t* src = static_cast<const int32_t*>(pcm_raw) + first;
        for (int s = threadIdx.x; s < S::kSpan; s += kThreads)
            span[s + S::kSkew * (s >> 6)] = (float)src[s] * (1.0f / 2147483648.0f);
    }
}

// Persistent workgroups, two per CU, as in k_rows_pruned.hip: every XCD owns a contiguous range of
// frames, the first ones are static, the rest is claimed a frame at a time from a per-XCD counter; the
// span of unit u + 1 streams into the span buffer as soon as every wave holds its points of unit u.  The
// twiddle tables are built once per workgroup.
template <int LOG2L, int FMT, bool S64, uint64_t SKIP>
__global__ __launch_bounds__(kThreads, 2) void rows_full_kernel(const void* __restrict__ pcm_raw, uint32_t stride_arg,
                                                                 uint64_t samples_per_clip,
                                                                 uint32_t frames_per_clip, uint64_t n_units,
                                                                 uint64_t units_per_xcd, const float* __restrict__ tw,
                                                                 const uint32_t* __restrict__ band_tbl, uint32_t nbands,
                                                                 uint32_t kmin, uint32_t kmax,
                                                                 uint32_t* __restrict__ claim_ctr,
                                                                 float* __restrict__ frames) {
    using S = Shape<LOG2L>;
    constexpr int L = S::L, N = S::N, R = S::R, WPW = S::WPW;
    constexpr uint32_t kUnitsPerFrame = 128 / S::UW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    // A window's power terms (round 5): band after band, every band starting on a bank of its own (BandTable::term_at) --
    // the lanes that add the bands' terms in bin order, one band per lane, never meet on a bank (by bin number they did four
    // at a time: most of the 35.6 % SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of round 4).  Where each of a lane's 64 bins
    // goes is a table in LDS (bins no band reads: the lane's dump word behind the last band); the windows of a wave lie an
    // ODD number of words apart: the 16 lanes of a window store bins two apart, the neighbour window fills the banks between.
    const uint32_t term_end = band_tbl[8 * nbands];
    const uint32_t pitch = term_pitch(term_end);
    const uint32_t stride = S64 ? (uint32_t)kStride : stride_arg;
    // LDS: [span][cross-lane twiddles: 64 rows][split-pass twiddles][claim slot][where the bins go][per wave: transpose pass / power terms]
    float* span = smem;
    float* ctw = smem + S::span_dw(stride, S64);
    // split-pass twiddles in consumption order: entry e = (slot q, pair u, half) of lane r at [e * L + r]
    // split-pass twiddles: W^ka of pair p = (slot q, pair u) of lane r at [p * L + r]; the partner bin's W^(N - ka) is
    // (-re, im) of it, exactly (make_twiddles builds the table by that symmetry) -- until round 5 it had its own entry
    float2* split_tw = reinterpret_cast<float2*>(ctw + 64 * S::kTwRowDw);
    uint32_t* claim_slot = reinterpret_cast<uint32_t*>(split_tw + 32 * L);
    uint16_t* term_at = reinterpret_cast<uint16_t*>(claim_slot + 4);          // entry e of lane r at [e * L + r], as split_tw
    const uint32_t wave_dw = wave_words<LOG2L>(term_end);
    float* tbuf = reinterpret_cast<float*>(term_at + 64 * L) + wave * wave_dw;
    float* vbuf = tbuf;   // the power terms reuse the wave's transpose area after the last pass

    // workgroup b runs on XCD b % 8 (observed dispatch order; speed only, see k_rows_pruned.hip)
    const uint32_t wg_per_xcd = gridDim.x >> 3;
    const uint64_t xcd_begin = (uint64_t)(blockIdx.x & 7) * units_per_xcd;     // a whole number of frames
    const uint64_t xcd_end = xcd_begin + units_per_xcd < n_units ? xcd_begin + units_per_xcd : n_units;
    uint64_t unit = xcd_begin + kUnitsPerFrame * (uint64_t)(blockIdx.x >> 3);
    if (unit >= xcd_end) return;
    auto span_start = [&](uint64_t u) {
        const uint32_t frame = (uint32_t)(u / kUnitsPerFrame);          // the launcher keeps unit numbers below 2^31
        const uint32_t part = (uint32_t)(u % kUnitsPerFrame);
        const uint32_t clip = frame / frames_per_clip;
        const uint32_t fi = frame - clip * frames_per_clip;
        return (uint64_t)clip * samples_per_clip + (uint64_t)(fi * 128 + part * S::UW) * stride;
    };
    uint32_t* my_ctr = claim_ctr + (blockIdx.x & 7);
    const bool claimer = threadIdx.x == 0;
    uint32_t claimed = 0;

    // ---- once per workgroup: first span, twiddle tables ----------------------------------------------
    span_to_lds<LOG2L, FMT, S64>(pcm_raw, span_start(unit), span, wave, lane, stride);
    // cross-lane stage t = 1..log2 L (overall stage s = 6 + t): butterfly jj of row k64 uses
    // W_(2^s)^(k64 + 64 jj) = tw[(k64 + 64 jj) << (LOG2W - s)]
    for (int i = threadIdx.x; i < 64 * (L - 1); i += kThreads) {
        const int row = i / (L - 1), e = i % (L - 1);
        int t = 1;
        while ((1 << t) - 1 <= e) ++t;
        const int jj = e - ((1 << (t - 1)) - 1);
        const uint32_t ti = (uint32_t)(row + 64 * jj) << (S::LOG2W - 6 - t);
        ctw[row * S::kTwRowDw + 2 * e] = tw[ti];
        ctw[row * S::kTwRowDw + 2 * e + 1] = tw[N + ti];
    }
    // every lane meets the same 64 bins in every unit: W^k of each, laid out the way the lanes read them
    // (conflict-free, no index arithmetic in the loop).  Bin numbering as in the split pass below.
    for (int i = threadIdx.x; i < 32 * L; i += kThreads) {
        const int rr = i % L, u = (i / L) % L, q = (i / L) / L;
        const int slot = rr * (R / 2) + q;
        int ka = slot + 64 * u;
        if (slot == 0 && u >= L / 2) ka = 32 + 64 * (u - L / 2);
        split_tw[i] = make_float2(tw[ka], tw[N + ka]);
        // pair 0 of slot 0 is (bin 0, bin N / 2): bin 0 takes no twiddle (DC and Nyquist, `dc` below), so the entry holds what
        // gives its partner W^(N/2) after the change of sign
        if (slot == 0 && u == 0) split_tw[i] = make_float2(-tw[N / 2], tw[N + N / 2]);
    }
    {
        // where the same 64 bins' power terms go; the bands' edges and first words pass through the (still unused) wave areas
        uint32_t* edge = reinterpret_cast<uint32_t*>(term_at + 64 * L);        // [lo | hi | first word][nbands]
        for (uint32_t i = threadIdx.x; i < nbands; i += kThreads) {
            edge[i] = band_tbl[i];
            edge[nbands + i] = band_tbl[nbands + i];
            edge[2 * nbands + i] = band_tbl[7 * nbands + i];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 64 * L; i += kThreads) {
            const int rr = i % L, e = i / L, half = e & 1, u = (e >> 1) % L, q = (e >> 1) / L;
            const int slot = rr * (R / 2) + q;
            int ka = slot + 64 * u;
            if (slot == 0 && u >= L / 2) ka = 32 + 64 * (u - L / 2);
            int kb = N - ka;
            if (slot == 0 && u == 0) kb = N / 2;
            const uint32_t k = (uint32_t)(half ? kb : ka);
            uint32_t at = term_end + (uint32_t)rr;                             // the lane's dump word
            if (!(half && kb == ka))
                for (uint32_t b = 0; b < nbands; ++b)
                    if (k >= edge[b] && k < edge[nbands + b]) at = edge[2 * nbands + b] + (k - edge[b]);
            term_at[i] = (uint16_t)at;
        }
        // (the loop below starts with a barrier: nobody writes a wave area before every thread is through here)
    }

    const int wl = lane / L, r = lane % L;          // window of the wave, lane of the window
    // The row of a pass this lane RECEIVES (and with it the bins, twiddles and term addresses of everything behind the
    // transpose): its own number -- at eight lanes per window the halves of windows 1 and 2 (mod 4) trade rows, which puts the
    // sixteen lanes of every ds_read_b128 lane group ({0-3, 12-15, 20-27}, ...: half of one window, half of each of the next
    // two, half of the fourth) on sixteen different 16-byte slots, as in k_rows_pruned.hip (same rows of 5 slots, same
    // window pitch of 4 slots mod 8 that the ds_write_b64 groups need).  What a lane CONTRIBUTES (its points, its column of
    // every row) stays with its own number.
#ifdef LBAD_EXP_RECV_NOSWAP
    const int r_recv = r;
#else
    const int r_recv = L == 8 ? r ^ ((((wl + 1) >> 1) & 1) << 2) : r;
#endif
    const float inv_norm = 1.0f / (float)(S::W / 4);
    float* tcol = tbuf + wl * S::kWinDw + 2 * r;                      // this lane's column of its window's pass
    const float* trow = tbuf + wl * S::kWinDw + r_recv * S::kRowDw;   // the row this lane receives

    for (;;) {
    // ---- A: this unit's span has landed (own loads: vmcnt, the other waves': barrier) -------------------
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    const uint32_t part = (uint32_t)(unit % kUnitsPerFrame);
    if (part == kUnitsPerFrame - 1 && claimer) claim_slot[0] = claimed;    // claimed at part 0 of this frame
    __syncthreads();

    // ---- B: 64 points of this lane; once every wave has its points the span buffer is free -----------
    cplx x[64];
    {
        // window w of the workgroup starts at sample stride w (stride 64: at float (64 + kSkew) w of the skewed copy)
        const int w = wave * WPW + wl;
        load_points64<LOG2L, S64, 0>(x, span + (S64 ? (64 + S::kSkew) : (int)stride) * w + 2 * r);
    }
    const uint64_t next = part == kUnitsPerFrame - 1
                              ? xcd_begin + kUnitsPerFrame * ((uint64_t)wg_per_xcd + claim_slot[0]) : unit + 1;
    __syncthreads();
    if (next < xcd_end) span_to_lds<LOG2L, FMT, S64>(pcm_raw, span_start(next), span, wave, lane, stride);
    if (part == 0 && claimer) claimed = atomicAdd(my_ctr, 1u);
    __builtin_amdgcn_s_setprio(0);     // arithmetic-heavy phase: let the co-resident wave's LDS work go first
    stage_blocks<1, 0>(x);
    stage_blocks<2, 0>(x);
    stage_blocks<3, 0>(x);
    stage_blocks<4, 0>(x);
    stage_blocks<5, 0>(x);
    stage_blocks<6, 0>(x);
    __builtin_amdgcn_s_setprio(3);

    // ---- C: log2 L cross-lane stages, R passes; pairs of passes end in the split pass ---------------
    // (bin numbers, twiddle addresses and store predicates are loop-invariant per lane; laundering r keeps
    // the compiler from hoisting a hundred of them out of the persistent loop and spilling them)
    int r_now = r_recv;
    asm volatile("" : "+v"(r_now));
    float pw[R / 2][2 * L];
    auto run_pass = [&](auto pass_tag, cplx (&y)[L], int row) {
        constexpr int P = decltype(pass_tag)::value;
        store_pass<LOG2L, P, 0>(x, tcol);
#pragma unroll
        for (int i = 0; i < L; i += 2) {
            const float4 q = *reinterpret_cast<const float4*>(trow + 2 * i);
            y[brevL<L>(i)] = mk(q.x, q.y);
            y[brevL<L>(i + 1)] = mk(q.z, q.w);
        }
        cross_fft<L>(y, ctw + row * S::kTwRowDw);
    };
    // power term of bin k from A = Z[k], B = Z[N - k] (LBAudioDetective.m:373-396 after the vDSP packing)
    const float2* my_tw = split_tw + r_now;
    // Written on (re, im) pairs (round 5): a + conj(b) and a - conj(b) as one packed add each; the two nested fmas of re
    // and of im as two v_pk_fma_f32 that take the twiddle pair as it comes from LDS -- op_sel picks wi for both halves
    // in the first and wr in the second, where the second operand's halves swap (di, dr) and neg_hi / neg_lo puts the
    // sign on wr: on the high half for W^k, on the low half for the partner's W^(N - k) = (-wr, wi).  Per half the same
    // correctly rounded operations on the same operands as
    //   re = fma(wr, di, fma(wi, dr, sr)),  im = fma(-wr, dr, fma(wi, di, si)).
    const cplx inv2 = mk(inv_norm, inv_norm);
    uint32_t worst = 0;
    auto power = [&](cplx a, cplx b, int p, auto partner_tag, bool dc) -> float {   // (bins outside the bands: computed, never read)
        constexpr bool PARTNER = decltype(partner_tag)::value;
        const cplx wk = *reinterpret_cast<const cplx*>(my_tw + p * L);
        cplx sm, df, z1, z;
        asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(sm) : "v"(a), "v"(b));
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(df) : "v"(a), "v"(b));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(z1) : "v"(wk), "v"(df), "v"(sm));
        if constexpr (PARTNER)
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]" : "=v"(z) : "v"(wk), "v"(df), "v"(z1));
        else
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "=v"(z) : "v"(wk), "v"(df), "v"(z1));
        if (dc) {                                                       // DC and Nyquist share bin 0
            const float s2 = a.x + a.y, d2 = a.x - a.y;
            z = mk(s2 + s2, d2 + d2);
        }
        // "if (x > 0) x *= 1 / (W/4)" is min(x * 2^-n, x): the same single rounding for x > 0, x itself
        // otherwise (negative, zero of either sign, NaN)
        // (v_min_f32 by hand: behind inline assembly the compiler does not know that z is no signalling NaN and would
        // put a v_max_f32 z, z in front of every fminf)
        const cplx zs = z * inv2;
        cplx zn;
        asm("v_min_f32 %0, %1, %2" : "=v"(zn.x) : "v"(zs.x), "v"(z.x));
        asm("v_min_f32 %0, %1, %2" : "=v"(zn.y) : "v"(zs.y), "v"(z.y));
        const cplx sq = zn * zn;
        const float t = __fadd_rn(sq.x, sq.y);
        // LBAudioDetective.m:398-401 skips NaN / inf terms: t >= +0.0 unless it is one of them, so as an unsigned integer every
        // such t lies at or above the bits of +inf -- one v_max_u32 per term keeps watch, and the unit that has one puts its
        // terms right in front of the stores below (was: a class test and a select per term)
        worst = max(worst, __float_as_uint(t));
        return t;
    };
    auto slot_work = [&](auto q_tag) {
        constexpr int Q = decltype(q_tag)::value;
        const int slot = r_now * (R / 2) + Q;
        const int row_a = slot, row_b = slot == 0 ? 32 : 64 - slot;
        cplx ya[L], yb[L];
        run_pass(std::integral_constant<int, 2 * Q>{}, ya, row_a);
        run_pass(std::integral_constant<int, 2 * Q + 1>{}, yb, row_b);
        // Split pass by pairs: bin ka = row_a + 64 u and its partner kb = N - ka = row_b + 64 (L - 1 - u).
        // Slot 0 (rows 0 and 32 pair with themselves) re-indexes: pair 0 = (bin 0, bin N/2), pairs
        // 1..L/2-1 = (64 u, 64 (L - u)) of row 0, pairs L/2.. = (32 + 64 v, 32 + 64 (L - 1 - v)) of row 32.
        static_for<L>([&](auto u_tag) {
            constexpr int u = decltype(u_tag)::value;
            constexpr int e = (Q * L + u) * 2;
            cplx a = ya[u], b = yb[L - 1 - u];
            if constexpr (Q == 0) {
                if (slot == 0) {
                    if (u == 0) { b = ya[L / 2]; }
                    else if (u < L / 2) { b = ya[L - u]; }
                    else { a = yb[u - L / 2]; b = yb[L - 1 - (u - L / 2)]; }
                }
            }
            cplx a2 = b, b2 = a;
            bool dc = false;
            if (Q == 0 && u == 0) {
                if (slot == 0) { a2 = b; b2 = b; b = a; dc = true; }     // (bin 0 from Z[0] alone, bin N/2 from Z[N/2] alone)
            }
            pw[Q][2 * u] = 0.0f;
            pw[Q][2 * u + 1] = 0.0f;
            if constexpr (((SKIP >> e) & 1ull) == 0) pw[Q][2 * u] = power(a, b, Q * L + u, std::false_type{}, dc);
            if constexpr (((SKIP >> (e + 1)) & 1ull) == 0) pw[Q][2 * u + 1] = power(a2, b2, Q * L + u, std::true_type{}, false);
        });
        // pin the order: without this the scheduler sinks every split pass below the last transpose and
        // keeps the outputs of all passes alive at once (several hundred bytes of scratch)
#pragma unroll
        for (int i = 0; i < 2 * L; ++i) asm volatile("" : "+v"(pw[Q][i]));
        // 512- / 256-sample windows have 8 / 16 slots of 4 / 2 values: without a hard fence the scheduler runs the
        // transposes of several slots ahead of their split passes and spills
        if constexpr (L <= 4) __builtin_amdgcn_sched_barrier(0);
    };
    auto all_slots = [&](auto self, auto q_tag) {
        constexpr int Q = decltype(q_tag)::value;
        if constexpr (Q < R / 2) {
            slot_work(q_tag);
            self(self, std::integral_constant<int, Q + 1>{});
        }
    };
    all_slots(all_slots, std::integral_constant<int, 0>{});

    // ---- D: power terms -> LDS (every read of the last pass has returned: the values are in
    //         registers), band means in bin order -----------------------------------------------------
    {
        if (__builtin_expect(__any(worst >= 0x7F800000u), 0)) {
#pragma unroll
            for (int q = 0; q < R / 2; ++q)
#pragma unroll
                for (int i = 0; i < 2 * L; ++i) {
                    const float t = pw[q][i];
                    pw[q][i] = (t == t && fabsf(t) != INFINITY) ? t : 0.0f;
                }
        }
        // a power term goes where the table says: no bin arithmetic, predicates or divergent stores in the loop
        float* vwin = vbuf + wl * pitch;
        const uint16_t* my_at = term_at + r_now;
        static_for<(R / 2) * L>([&](auto p_tag) {
            constexpr int e = decltype(p_tag)::value * 2, q = decltype(p_tag)::value / L, u = decltype(p_tag)::value % L;
            if constexpr (((SKIP >> e) & 1ull) == 0) vwin[my_at[e * L]] = pw[q][2 * u];
            if constexpr (((SKIP >> (e + 1)) & 1ull) == 0) vwin[my_at[(e + 1) * L]] = pw[q][2 * u + 1];
        });
    }
    // (wave-local: LDS operations of one wave execute in order)
    // task t = lane + 64 i -> window t / nbands of the wave, band t % nbands; rows of a wave's windows are
    // consecutive, so the task's float sits at rows[t]
    auto band_mean = [&](uint32_t t) -> float {
        float p = 0.0f;
        float div = 1.0f;
        if (t < WPW * nbands) {
            const uint32_t ww = t / nbands, band = t - ww * nbands;
            const uint32_t lo = band_tbl[band], hi = band_tbl[nbands + band];
            div = __uint_as_float(band_tbl[2 * nbands + band]);
            // whole batches of eight terms need no per-term mask (a lane leaves the loop when its band runs out);
            // the last w % 8 terms are read from the lane's own offset -- what lies behind a band (another band's
            // terms, padding, the scratch words) is read and selected away, never added
            const uint32_t width = hi > lo ? hi - lo : 0;
            const float* vb = vbuf + ww * pitch + band_tbl[7 * nbands + band];
            const uint32_t full = width >> 3, rem = width & 7;
            for (uint32_t b = 0; b < full; ++b) {
                float v[8];
#pragma unroll
                for (uint32_t q = 0; q < 8; ++q) v[q] = vb[8 * b + q];
#pragma unroll
                for (uint32_t q = 0; q < 8; ++q) p = __fadd_rn(p, v[q]);
            }
            if (rem) {
                float v[7];
#pragma unroll
                for (uint32_t q = 0; q < 7; ++q) v[q] = vb[8 * full + q];
#pragma unroll
                for (uint32_t q = 0; q < 7; ++q) p = __fadd_rn(p, q < rem ? v[q] : 0.0f);
            }
        }
        return __fdiv_rn(p, div);
    };
    // (stored at once by a plain loop over the wave's tasks: holding the means back until the next unit's loads are
    // out, as round 1 did, costs WPW registers through the FFT -- spills at 16 / 32 windows per wave -- and gains nothing)
    float* rows = frames + ((unit / kUnitsPerFrame) * 128 + part * S::UW + wave * WPW) * nbands;
    for (uint32_t t = lane; t < WPW * nbands; t += 64) rows[t] = band_mean(t);
    if (next >= xcd_end) break;
    unit = next;
    }
}

template <int LOG2L> size_t lds_bytes(uint32_t term_end, uint32_t stride) {
    using S = Shape<LOG2L>;
    return ((size_t)S::span_dw(stride, stride == (uint32_t)kStride) + 64 * S::kTwRowDw + 2 * 32 * (size_t)S::L + 4 +
            (size_t)kWaves * wave_words<LOG2L>(term_end)) * sizeof(float) + 64 * (size_t)S::L * sizeof(uint16_t);
}

template <int LOG2L, int FMT, bool S64, uint64_t SKIP = 0>
hipError_t launch_full_fmt(const Plan& plan, const void* d_pcm, uint64_t n_frames, uint64_t samples_per_clip,
                           uint32_t frames_per_clip, float* d_frames, hipStream_t stream) {
    using S = Shape<LOG2L>;
    const size_t lds = lds_bytes<LOG2L>(plan.table.term_end, plan.stride);
    static PerDevice attr;
    if (attr.changed(lds)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rows_full_kernel<LOG2L, FMT, S64, SKIP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    // work is claimed by frames: every XCD's range is a whole number of frames
    constexpr uint64_t U = 128 / S::UW;
    const uint64_t n_units = n_frames * U, units_per_xcd = U * ((n_frames + 7) / 8);
    if (units_per_xcd * 8 > 0x7fffffffull) return hipErrorInvalidValue;
    const int n_cu = device_cu_count();
    uint64_t wg_per_xcd = ((uint64_t)n_cu * 2 + 7) / 8;          // two persistent workgroups per CU
    if (wg_per_xcd > units_per_xcd / U) wg_per_xcd = units_per_xcd / U;
    hipError_t e = hipMemsetAsync(plan.d_claim, 0, 8 * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((rows_full_kernel<LOG2L, FMT, S64, SKIP>), dim3((uint32_t)(wg_per_xcd * 8)), dim3(kThreads), lds, stream,
                       d_pcm, plan.stride, samples_per_clip, frames_per_clip, n_units, units_per_xcd, plan.d_tw, plan.d_bands,
                       plan.bands, plan.table.kmin, plan.table.kmax, plan.d_claim, d_frames);
    return hipGetLastError();
}

template <int LOG2L>
hipError_t launch_full(const Plan& plan, const void* d_pcm, uint32_t fmt, uint64_t n_frames, uint64_t samples_per_clip,
                       uint32_t frames_per_clip, float* d_frames, hipStream_t stream) {
    // 2048-sample windows, float32: the instance without the terms the default table never reads, when this table does not either
    const bool lean = LOG2L == 4 && fmt == 0 && (plan.table.unread_terms16 & kDefaultUnread16) == kDefaultUnread16;
    constexpr uint64_t kLean = LOG2L == 4 ? kDefaultUnread16 : 0ull;
    if (plan.stride != (uint32_t)kStride) {                // any other stride: float32 input only (rows_full_supported_fmt)
        if (fmt != 0) return hipErrorInvalidValue;
        if (lean) return launch_full_fmt<LOG2L, 0, false, kLean>(plan, d_pcm, n_frames, samples_per_clip, frames_per_clip, d_frames, stream);
        return launch_full_fmt<LOG2L, 0, false>(plan, d_pcm, n_frames, samples_per_clip, frames_per_clip, d_frames, stream);
    }
    if (lean) return launch_full_fmt<LOG2L, 0, true, kLean>(plan, d_pcm, n_frames, samples_per_clip, frames_per_clip, d_frames, stream);
    switch (fmt) {
        case 0: return launch_full_fmt<LOG2L, 0, true>(plan, d_pcm, n_frames, samples_per_clip, frames_per_clip, d_frames, stream);
        case 1: return launch_full_fmt<LOG2L, 1, true>(plan, d_pcm, n_frames, samples_per_clip, frames_per_clip, d_frames, stream);
        case 2: return launch_full_fmt<LOG2L, 2, true>(plan, d_pcm, n_frames, samples_per_clip, frames_per_clip, d_frames, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace

// 1024- and 2048-sample windows at the reference's stride; the compile-time W_64 table must be
// bit-identical to the run-time master table and the LDS budget must allow two workgroups per CU
bool rows_full_supported(const Plan& p) {
    // even strides: a lane reads its points as aligned sample pairs (ds_read_b64)
    if (p.stride == 0 || (p.stride & 1u) || p.stride > 1024 || p.bands > 64 || p.bands == 0) return false;
    if (p.window != 256 && p.window != 512 && p.window != 1024 && p.window != 2048) return false;
    if (p.table.kmax <= p.table.kmin) return false;
    std::vector<float> re, im;
    make_twiddles(p.window, re, im);
    const uint32_t step = p.window / 64;
    for (int t = 0; t < 32; ++t)
        if (re[step * t] != kTw64Re[t] || im[step * t] != kTw64Im[t]) return false;
    const size_t lds = p.window == 256    ? lds_bytes<1>(p.table.term_end, p.stride)
                       : p.window == 512  ? lds_bytes<2>(p.table.term_end, p.stride)
                       : p.window == 1024 ? lds_bytes<3>(p.table.term_end, p.stride)
                                          : lds_bytes<4>(p.table.term_end, p.stride);
    return lds <= 80 * 1024;
}

// strides other than 64 take float32 input (integer PCM at those strides runs on the generic kernel)
bool rows_full_supported_fmt(const Plan& p, uint32_t fmt) { return p.stride == (uint32_t)kStride || fmt == 0; }

hipError_t launch_rows_full(const Plan& plan, const void* d_pcm, uint32_t fmt, uint64_t n_clips, uint64_t samples_per_clip,
                            uint32_t frames_per_clip, float* d_frames, hipStream_t stream) {
    const uint64_t n_frames = n_clips * frames_per_clip;
    if (n_frames == 0) return hipSuccess;
    if (n_frames > 0x7ffffffull) return hipErrorInvalidValue;
    if (plan.window == 256) return launch_full<1>(plan, d_pcm, fmt, n_frames, samples_per_clip, frames_per_clip, d_frames, stream);
    if (plan.window == 512) return launch_full<2>(plan, d_pcm, fmt, n_frames, samples_per_clip, frames_per_clip, d_frames, stream);
    if (plan.window == 1024) return launch_full<3>(plan, d_pcm, fmt, n_frames, samples_per_clip, frames_per_clip, d_frames, stream);
    return launch_full<4>(plan, d_pcm, fmt, n_frames, samples_per_clip, frames_per_clip, d_frames, stream);
}

}  // namespace lbad
