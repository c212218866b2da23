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
//            stages 1..4 in registers, compile-time twiddles) -> the halves trade D4s (v_permlane32_swap)
//            and stage 5 leaves D5(g)[16 h + kk] in lane (n, h) -> stage 6 with the D5 of the previous step,
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
#include "internal.hpp"
#include "fft64_lane.hpp"

namespace lbad {
namespace {

using namespace lane64;

constexpr int kW = 4096;
constexpr int kN = kW / 2;
constexpr int kStride = 64;
constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;
constexpr int kRowDw = 68;                    // 32 complex + 16 B: lanes 0..15 of a b128 read hit disjoint banks
constexpr int kTDw = 64 * kRowDw;             // transpose of one window (one wave)
constexpr int kCrossTw = 27;                  // cross-stage twiddles a lane uses: 1 + 2 + 4 + 8 + 12
constexpr int kQ = 6;                         // low outputs per row: bins a + 64 q < 384
constexpr int kMaxBin = 64 * kQ;
constexpr int kPowerDw = kMaxBin + 64;        // power terms of a window + one dummy word per lane
constexpr int kLdsDw = kWaves * kTDw + 64 * kRowDw + kQ * 64 * 2 + 2 * 32 * 2;
constexpr int kLdsBytes = kLdsDw * 4;         // 160 256 B: one workgroup per CU
static_assert(kLdsBytes <= 160 * 1024, "LDS budget");
static_assert(kPowerDw <= kTDw, "power terms reuse the transpose area");

__device__ __forceinline__ constexpr int brev4(int v) { return ((v & 1) << 3) | ((v & 2) << 1) | ((v & 4) >> 1) | ((v & 8) >> 3); }
__device__ __forceinline__ constexpr int brev5(int v) {
    return ((v & 1) << 4) | ((v & 2) << 2) | (v & 4) | ((v & 8) >> 2) | ((v & 16) >> 4);
}

// 16-point DIT stages on registers (same butterflies as lane64::stage_blocks, 16 slots)
template <int S, int BASE, int J>
__device__ __forceinline__ void st16_j(cplx (&x)[16]) {
    constexpr int half = 1 << (S - 1);
    if constexpr (J < half) {
        bfly<J*(64 >> S)>(x[BASE + J], x[BASE + J + half]);
        st16_j<S, BASE, J + 1>(x);
    }
}
template <int S, int BASE>
__device__ __forceinline__ void st16(cplx (&x)[16]) {
    if constexpr (BASE < 16) {
        st16_j<S, BASE, 0>(x);
        st16<S, BASE + (1 << S)>(x);
    }
}

// one complex point = two consecutive samples
template <int FMT>
__device__ __forceinline__ cplx load_point(const void* p, int64_t idx) {
    if constexpr (FMT == 0) {
        return *reinterpret_cast<const f32x2*>(static_cast<const float*>(p) + 2 * idx);
    } else if constexpr (FMT == 1) {
        const short2 s = *reinterpret_cast<const short2*>(static_cast<const int16_t*>(p) + 2 * idx);
        return mk((float)s.x * (1.0f / 32768.0f), (float)s.y * (1.0f / 32768.0f));
    } else {
        const int2 s = *reinterpret_cast<const int2*>(static_cast<const int32_t*>(p) + 2 * idx);
        return mk((float)s.x * (1.0f / 2147483648.0f), (float)s.y * (1.0f / 2147483648.0f));
    }
}

template <int FMT, int T>
__device__ __forceinline__ void load16(cplx (&x)[16], const void* p, int64_t idx) {
    if constexpr (T < 16) {
        x[T] = load_point<FMT>(p, idx + 128 * brev4(T));     // slot T holds point m' = brev4(T)
        load16<FMT, T + 1>(x, p, idx);
    }
}

__device__ __forceinline__ float swap_halves_u(float t, float x, float& v_out) {
    // v_permlane32_swap vdst = t, src = x: lanes 32..63 of t trade with lanes 0..31 of x.  With t a copy of x
    // every lane ends up with (t, x) = (the lower lane's value, the upper lane's value).
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(x), false, false);
    v_out = __uint_as_float(r[1]);
    return __uint_as_float(r[0]);
}

__device__ __forceinline__ float dpp_pair_swap(float v) {      // lane l <-> lane l ^ 1
    return __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
}

__device__ __forceinline__ void wave_sync() {
    // LDS operations of one wave execute in order; this only stops the compiler from moving them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ cplx msub(cplx u, float wr, float wi, cplx v) {     // u - w v
    return fma2(mk(-wr, -wr), v, fma2(mk(wi, -wi), v.yx, u));
}

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
                                                                  uint32_t kmin, uint32_t kmax,
                                                                  uint32_t* __restrict__ claim_ctr,
                                                                  float* __restrict__ frames) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    float* tbuf = smem + wave * kTDw;                       // this wave's transpose / power terms
    float* ctw = smem + kWaves * kTDw;                      // [lane][27 complex], row pitch kRowDw
    float2* stw = reinterpret_cast<float2*>(ctw + 64 * kRowDw);   // [q][lane]
    float2* p1tw = stw + kQ * 64;                           // [h][32]: 16 stage-5 (sign folded) + 16 stage-6 twiddles

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
        ctw[l * kRowDw + 2 * e] = tw[ti];
        ctw[l * kRowDw + 2 * e + 1] = tw[kN + ti];
    }
    for (int i = threadIdx.x; i < kQ * 64; i += kThreads) {
        const int q = i / 64, l = i % 64;
        const uint32_t k = (uint32_t)(row_of_lane(l) + 64 * q);     // < 2048
        stw[i] = make_float2(tw[k], tw[kN + k]);
    }
    for (int i = threadIdx.x; i < 64; i += kThreads) {
        const int h = i >> 5, e = i & 31;
        float wr, wi;
        if (e < 16) {                                        // stage 5: W_32^kk, negated for the "-" half
            wr = tw[e * (kW / 32)];
            wi = tw[kN + e * (kW / 32)];
            if (h) { wr = -wr; wi = -wi; }
        } else {                                             // stage 6: W_64^(16 h + kk)
            wr = tw[(16 * h + e - 16) * (kW / 64)];
            wi = tw[kN + (16 * h + e - 16) * (kW / 64)];
        }
        p1tw[i] = make_float2(wr, wi);
    }
    __syncthreads();

    const int n = lane & 31, h = lane >> 5;
    const int my_row = row_of_lane(lane);
    const bool special = lane < 2;                           // rows 0 and 32 pair with themselves
    const float inv_norm = 1.0f / (float)(kW / 4);
    const float2* my_p1 = p1tw + 32 * h;
    float* my_col = tbuf + n * 2;                            // column n of the transpose (row pitch kRowDw)
    const float* my_trow = tbuf + lane * kRowDw;             // the row this lane transforms (stored by destination lane)
    const float* my_ctw = ctw + lane * kRowDw;
    // destination lane of row k (the inverse of row_of_lane): 0 -> 0, 32 -> 1, k < 32 -> 2 k, else 2 (64 - k) + 1.
    // Lane (n, h) emits rows 16 h + kk -> lane 32 h + 2 kk, and rows 32 + 16 h + kk -> lane (h ? 33 : 65) - 2 kk
    // (row 32 itself, h = 0 and kk = 0, -> lane 1).
    float* col_p = my_col + 32 * h * kRowDw;
    float* col_m = my_col + (h ? 33 : 65) * kRowDw;
    float* col_m0 = my_col + (h ? 33 : 1) * kRowDw;
    // which of this lane's six low bins a band reads, and where their power terms go
    uint32_t need = 0;
#pragma unroll
    for (int q = 0; q < kQ; ++q) {
        const uint32_t k = (uint32_t)(my_row + 64 * q);
        if (k >= kmin && k < kmax && k != 0) need |= 1u << q;
    }
    float* vbuf = tbuf;
    float* dummy = tbuf + kMaxBin + lane;
    uint32_t b_lo = 0, b_hi = 0;
    float b_div = 1.0f;
    if ((uint32_t)lane < nbands) {
        b_lo = band_tbl[lane];
        b_hi = band_tbl[nbands + lane];
        b_div = __uint_as_float(band_tbl[2 * nbands + lane]);
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
    auto d5_block = [&](const cplx (&x_in)[16], cplx (&out)[16]) {
        cplx x[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) x[t] = x_in[t];
        st16<1, 0>(x);
        st16<2, 0>(x);
        st16<3, 0>(x);
        st16<4, 0>(x);
        // the halves trade their D4s: u = D4(g) from lane (n, 0), v = D4(g + 64) from lane (n, 1)
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float vx, vy;
            const float ux = swap_halves_u(x[kk].x, x[kk].x, vx);
            const float uy = swap_halves_u(x[kk].y, x[kk].y, vy);
            const cplx u = mk(ux, uy), v = mk(vx, vy);
            const float2 w = my_p1[kk];                                  // +-W_32^kk
            out[kk] = madd(u, w.x, w.y, v);
        }
    };

    uint32_t frame = claim();
    while (frame < f_end) {
        const uint32_t next_frame = claim();                             // in flight for a whole frame
        const uint32_t clip = frame / frames_per_clip;
        const uint32_t fi = frame - clip * frames_per_clip;
        // complex point 0 of the frame; lane (n, h) reads points g + 64 h + 128 m'
        const int64_t c0 = (int64_t)(((uint64_t)clip * samples_per_clip + (uint64_t)fi * 128 * kStride) >> 1) + n + 64 * h;
        float* out_row = frames + (uint64_t)frame * 128 * nbands + lane;

        cplx xa[16], xb[16], P[16], Nw[16];
        load16<FMT, 0>(xa, pcm, c0);
        load16<FMT, 0>(xb, pcm, c0 + 32);
        d5_block(xa, P);

        // one window: x holds the points of block `step`, xn receives those of block `step + 1`
        auto window_step = [&](int step, cplx (&x)[16], cplx (&xn)[16], cplx (&Pp)[16], cplx (&Nn)[16]) {
            if (step < 128) load16<FMT, 0>(xn, pcm, c0 + 32 * (step + 1));
            d5_block(x, Nn);
            // ---- stage 6 and the transpose: rows k = 16 h + kk and k + 32, column n; a row is stored at
            //      the slot of the lane that will transform it (col_p / col_m, affine in kk) ------------------
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const float2 w = my_p1[16 + kk];                         // W_64^(16 h + kk)
                const cplx ep = madd(Pp[kk], w.x, w.y, Nn[kk]);
                const cplx em = msub(Pp[kk], w.x, w.y, Nn[kk]);
                *(lds_vf32x2*)(col_p + 2 * kk * kRowDw) = ep;
                *(lds_vf32x2*)(kk == 0 ? col_m0 : col_m - 2 * kk * kRowDw) = em;
            }
            wave_sync();

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
                    y[b] = madd(u, w0.x, w0.y, v);
                    y[b + 1] = msub(u, w0.x, w0.y, v);
                }
            }
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const f32x2 w = *reinterpret_cast<const f32x2*>(my_ctw + 2 * (1 + jj));
#pragma unroll
                for (int b = 0; b < 32; b += 4) {
                    const cplx u = y[b + jj], v = y[b + jj + 2];
                    y[b + jj] = madd(u, w.x, w.y, v);
                    y[b + jj + 2] = msub(u, w.x, w.y, v);
                }
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const f32x2 w = *reinterpret_cast<const f32x2*>(my_ctw + 2 * (3 + jj));
#pragma unroll
                for (int b = 0; b < 32; b += 8) {
                    const cplx u = y[b + jj], v = y[b + jj + 4];
                    y[b + jj] = madd(u, w.x, w.y, v);
                    y[b + jj + 4] = msub(u, w.x, w.y, v);
                }
            }
            // cross stage 4, pruned: per 16-block the outputs p in {0..5} ("+") and {10..15} ("-")
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const f32x2 w = *reinterpret_cast<const f32x2*>(my_ctw + 2 * (7 + jj));
#pragma unroll
                for (int b = 0; b < 32; b += 16) {
                    const cplx u = y[b + jj], v = y[b + jj + 8];
                    if (jj < 6) y[b + jj] = madd(u, w.x, w.y, v);
                    if (jj >= 2) y[b + jj + 8] = msub(u, w.x, w.y, v);
                }
            }
            // cross stage 5, pruned: outputs q in {0..5} ("+") and {26..31} ("-" of pairs 10..15)
            cplx lo[kQ], hi[kQ];
#pragma unroll
            for (int q = 0; q < kQ; ++q) {
                const f32x2 wa = *reinterpret_cast<const f32x2*>(my_ctw + 2 * (15 + q));
                lo[q] = madd(y[q], wa.x, wa.y, y[q + 16]);
                const f32x2 wb = *reinterpret_cast<const f32x2*>(my_ctw + 2 * (21 + q));
                hi[q] = msub(y[10 + q], wb.x, wb.y, y[26 + q]);
            }
            // ---- split pass: bin k = row + 64 q needs Z[N - k] = output 31 - q of the partner row (pair 0:
            //      output 31 - q of row 32 itself, output 32 - q of row 0 itself) --------------------------
            float pw[kQ];
#pragma unroll
            for (int q = 0; q < kQ; ++q) {
                const int j = kQ - 1 - q;                                // hi[j] = output 26 + j = 31 - q
                cplx b;
                b.x = dpp_pair_swap(hi[j].x);
                b.y = dpp_pair_swap(hi[j].y);
                const cplx own = (lane & 1) ? hi[j] : hi[j + 1 < kQ ? j + 1 : j];   // (bin 0 of row 0 is never read)
                if (special) b = own;
                const cplx a = lo[q];
                const float2 wk = stw[q * 64 + lane];
                const float sr = a.x + b.x, si = a.y - b.y;
                const float dr = a.x - b.x, di = a.y + b.y;
                float re = __fmaf_rn(wk.x, di, __fmaf_rn(wk.y, dr, sr));
                float im = __fmaf_rn(-wk.x, dr, __fmaf_rn(wk.y, di, si));
                // "if (x > 0) x /= W/4" is min(x * 2^-10, x): one rounding for x > 0, x itself otherwise
                re = fminf(__fmul_rn(re, inv_norm), re);
                im = fminf(__fmul_rn(im, inv_norm), im);
                pw[q] = __fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im));
            }
            wave_sync();                                                  // every row has been read
#pragma unroll
            for (int q = 0; q < kQ; ++q) {
                float* dst = (need >> q) & 1u ? vbuf + (my_row + 64 * q) : dummy;
                *dst = pw[q];
            }
            wave_sync();
            // ---- band means in bin order (LBAudioDetective.m:379-405) -------------------------------------
            float p = 0.0f;
            for (uint32_t k0 = b_lo; k0 < b_hi; k0 += 8) {
                float v[8];
#pragma unroll
                for (uint32_t q = 0; q < 8; ++q) v[q] = (k0 + q < b_hi) ? vbuf[k0 + q] : 0.0f;
#pragma unroll
                for (uint32_t q = 0; q < 8; ++q) v[q] = (v[q] == v[q] && fabsf(v[q]) != INFINITY) ? v[q] : 0.0f;
#pragma unroll
                for (uint32_t q = 0; q < 8; ++q) p = __fadd_rn(p, v[q]);
            }
            if ((uint32_t)lane < nbands) out_row[(uint64_t)(step - 1) * nbands] = __fdiv_rn(p, b_div);
            wave_sync();                                                  // the power terms are consumed
        };

        for (int step = 1; step <= 128; step += 2) {
            window_step(step, xb, xa, P, Nw);
            window_step(step + 1, xa, xb, Nw, P);
        }
        frame = next_frame;
    }
}

}  // namespace

bool rows_stream_supported(const Plan& p) {
    if (p.window != (uint32_t)kW || p.stride != (uint32_t)kStride || p.bands == 0 || p.bands > 64) return false;
    if (p.table.kmax <= p.table.kmin || p.table.kmin < 1 || p.table.kmax > (uint32_t)kMaxBin) return false;
    std::vector<float> re, im;
    make_twiddles(kW, re, im);
    for (int t = 0; t < 32; ++t)
        if (re[(kW / 64) * t] != kTw64Re[t] || im[(kW / 64) * t] != kTw64Im[t]) return false;
    return true;
}

template <int FMT>
static hipError_t launch_stream_fmt(const Plan& plan, const void* d_pcm, uint64_t n_frames, uint64_t samples_per_clip,
                                    uint32_t frames_per_clip, float* d_frames, hipStream_t stream) {
    static PerDevice attr;
    if (attr.changed(kLdsBytes)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rows_stream_kernel<FMT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        if (e != hipSuccess) return e;
    }
    const uint32_t frames_per_xcd = (uint32_t)((n_frames + 7) / 8);
    const uint32_t waves_per_xcd = (uint32_t)device_cu_count() / 8 * kWaves;
    // no more workgroups than an XCD has frames to hand out (a wave that finds no frame exits at once)
    uint32_t wg_per_xcd = (uint32_t)device_cu_count() / 8;
    while (wg_per_xcd > 1 && (uint64_t)(wg_per_xcd - 1) * kWaves >= frames_per_xcd) --wg_per_xcd;
    (void)waves_per_xcd;
    hipError_t e = hipMemsetAsync(plan.d_claim, 0, 8 * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(rows_stream_kernel<FMT>, dim3(wg_per_xcd * 8), dim3(kThreads), kLdsBytes, stream, d_pcm,
                       samples_per_clip, frames_per_clip, (uint32_t)n_frames, frames_per_xcd, plan.d_tw, plan.d_bands,
                       plan.bands, plan.table.kmin, plan.table.kmax, plan.d_claim, d_frames);
    return hipGetLastError();
}

hipError_t launch_rows_stream(const Plan& plan, const void* d_pcm, uint32_t fmt, uint64_t n_clips, uint64_t samples_per_clip,
                              uint32_t frames_per_clip, float* d_frames, hipStream_t stream) {
    const uint64_t n_frames = n_clips * frames_per_clip;
    if (n_frames == 0) return hipSuccess;
    if (n_frames > 0x7fffffffull) return hipErrorInvalidValue;
    switch (fmt) {
        case 0: return launch_stream_fmt<0>(plan, d_pcm, n_frames, samples_per_clip, frames_per_clip, d_frames, stream);
        case 1: return launch_stream_fmt<1>(plan, d_pcm, n_frames, samples_per_clip, frames_per_clip, d_frames, stream);
        case 2: return launch_stream_fmt<2>(plan, d_pcm, n_frames, samples_per_clip, frames_per_clip, d_frames, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace lbad
