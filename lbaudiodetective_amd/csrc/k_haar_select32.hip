// k_haar_select32.hip -- specialised stage 2 for 128 x 32 frames (and, since round 3, 128 x 16 and 128 x 64):
// 2-D Haar + ranked top-K signs.
//
// Same results as k_haar_select.hip (LBAudioDetectiveFrame.m:113-153,165-191 and the truncating
// copy at LBAudioDetective.m:326-328), restructured for the default frame shape:
//
//   * row pass: COLS / 16 threads per row (one, two or four), 16 values each, Haar levels 1..4 in registers, the
//     remaining log2(COLS / 16) levels with lane exchanges; results go to LDS already transposed (column-major,
//     16-row chunks);
//   * column pass: eight threads per column, 16 values each, levels 1..4 in registers, levels
//     5..7 with three lane exchanges; every thread ends up owning 16 final coefficients whose
//     flat positions are known in closed form, so the select works on registers;
//   * select: bisect the |v| bit patterns for a threshold that leaves [keep, 128] candidates
//     (one workgroup reduction per step), then rank only the candidates by the composite key
//     (|v| bits, lower flat index first) and emit the sign pairs of ranks < keep.
//
// Every arithmetic operation is the reference's: x / sqrtf(n) pre-scale, (a +- b) / sqrtf(2)
// butterflies, correctly rounded divisions.
#include "internal.hpp"

#include <type_traits>

namespace lbad {
namespace {

constexpr uint32_t kCand = 128;     // candidates ranked exhaustively
constexpr int kChunkDw = 20;        // 16 floats + 4 pad: conflict-free ds_read_b128 across lanes

// x / d for the Haar's three constant divisors without the ~11-instruction IEEE division sequence:
//     q0 = x * r;  e = fma(-d, q0, x);  q = fma(e, r, q0)        with r = RN(1 / d).
// tools/verify_const_div.c checks ALL 2^32 inputs for d = sqrtf(2), sqrtf(32), sqrtf(128): q equals the
// correctly rounded quotient bit for bit whenever 2^-105 <= |x| < inf (and for +0; -0 gives +0, which
// no later step can tell apart).  The guard records what a line met and the caller redoes the whole
// line with true divisions when a lane saw anything else:
//   * every dividend: min over t = 2 |x|bits - 1 (one shift-add per value, one min3 per pair); zero wraps
//     to 0xFFFFFFFF and never trips it, anything in (0, 2^-100) does;
//   * the 16 values a line starts from: max over |x|bits <= 2^126.  Later dividends are sums and
//     differences of quotients, at most 4x the largest input after four levels, so they stay finite.
constexpr uint32_t kFastDivLo = 0x0D800000u;   // 2^-100
constexpr uint32_t kFastDivHi = 0x7E800000u;   // 2^126

struct DivGuard {
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    __device__ __forceinline__ void dividends(float a, float b) {
        lo = min(lo, min((__float_as_uint(a) << 1) - 1u, (__float_as_uint(b) << 1) - 1u));
    }
    __device__ __forceinline__ void inputs(float a, float b) {
        hi = max(hi, max(__float_as_uint(a) & 0x7fffffffu, __float_as_uint(b) & 0x7fffffffu));
    }
    __device__ __forceinline__ bool bad() const { return lo < 2u * kFastDivLo - 1u || hi > kFastDivHi; }
};

template <bool FAST>
__device__ __forceinline__ float div_c(float x, float d, float r) {
    if constexpr (FAST) {
        const float q0 = __fmul_rn(x, r);
        const float e = __fmaf_rn(-d, q0, x);
        return __fmaf_rn(e, r, q0);
    } else {
        return __fdiv_rn(x, d);
    }
}

constexpr uint32_t kHistBuckets = 2048;   // |v| bits >> 20

// inclusive prefix sum over the 64 lanes of a wave (full EXEC): four row shifts, two row broadcasts (gfx9 DPP)
__device__ __forceinline__ uint32_t wave_prefix_sum(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111 /* row_shr:1 */, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112 /* row_shr:2 */, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114 /* row_shr:4 */, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118 /* row_shr:8 */, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142 /* row_bcast:15 */, 0xA, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143 /* row_bcast:31 */, 0xC, 0xF, false);
    return x;
}

template <int WAVES>
__device__ __forceinline__ uint32_t slots_sum(const uint32_t* slot) {
    uint32_t c = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) c += slot[w];
    return c;
}

template <int WAVES>
__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t* s_red, int parity) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    uint32_t* slot = s_red + 8 * parity;   // double-buffered: one barrier per reduction
    if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = v;
    __syncthreads();
    return slots_sum<WAVES>(slot);
}

// Most lines need far fewer checks (round 3).  A sum or difference of two floats a, b is 0 or at least
// min(ulp a, ulp b) > min(|a|, |b|) 2^-24, and a quotient by sqrt 2 loses another half binade; the pre-scale divides by
// at most 2^3.5.  So when every NON-ZERO input of a line is >= 2^-52, the dividends of levels 1 and 2 are 0 or
// >= 2^-79.5, 2^-104 -- inside the verified range without looking at them.  The light guard therefore checks the 16
// inputs (min over t as above against 2^-52; magnitudes with one v_max3_f32 per pair, which ignores NaN -- a NaN input
// reaches the level-4 average through every sum and is caught there) and the six dividends of levels 3 and 4.  A line
// that fails it is redone with the full guard, and with true divisions if that fails too.
constexpr uint32_t kLightLo = 0x25800000u;     // 2^-52
typedef float f2 __attribute__((ext_vector_type(2)));

struct LightGuard {
    uint32_t lo_in = 0xFFFFFFFFu, lo_4 = 0xFFFFFFFFu;
    float hi = 0.0f;
    __device__ __forceinline__ void levels34(f2 a, f2 b, f2 c) {
        auto t = [](float x) { return (__float_as_uint(x) << 1) - 1u; };
        lo_4 = min(min(min(t(a.x), t(a.y)), min(t(b.x), t(b.y))), min(t(c.x), t(c.y)));
    }
    __device__ __forceinline__ bool bad(float s4) const {
        return lo_in < 2u * kLightLo - 1u || lo_4 < 2u * kFastDivLo - 1u || !(hi <= __uint_as_float(kFastDivHi)) || s4 != s4;
    }
};

// levels 1..4 of a 16-value line held in registers, compiler-scheduled: FAST = the shortcut with every dividend guarded,
// otherwise true divisions.  On return d[0..7] = level-1 details, d[8..11] = level 2, d[12..13] = level 3, d[14] = level 4
// and the return value is the level-4 average that continues into the cross-lane levels.
template <bool FAST>
__device__ __forceinline__ float haar16_impl(const float (&in)[16], float (&d)[15], float root, float r_root,
                                             float root2, float r_root2, DivGuard& g) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        if constexpr (FAST) {
            g.inputs(in[i], in[i + 1]);
            g.dividends(in[i], in[i + 1]);
        }
        a[i] = div_c<FAST>(in[i], root, r_root);
        a[i + 1] = div_c<FAST>(in[i + 1], root, r_root);
    }
    // one butterfly: (x + y) / sqrt 2 and (x - y) / sqrt 2
    auto bfly = [&](float x, float y, float& sum, float& dif) {
        const float sp = __fadd_rn(x, y), sm = __fsub_rn(x, y);
        if constexpr (FAST) g.dividends(sp, sm);
        sum = div_c<FAST>(sp, root2, r_root2);
        dif = div_c<FAST>(sm, root2, r_root2);
    };
    float s1[8], s2[4], s3[2], s4;
#pragma unroll
    for (int i = 0; i < 8; ++i) bfly(a[2 * i], a[2 * i + 1], s1[i], d[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) bfly(s1[2 * i], s1[2 * i + 1], s2[i], d[8 + i]);
#pragma unroll
    for (int i = 0; i < 2; ++i) bfly(s2[2 * i], s2[2 * i + 1], s3[i], d[12 + i]);
    bfly(s3[0], s3[1], s4, d[14]);
    return s4;
}

// ---- the light tier, written out (round 3) ---------------------------------------------------------------------------
// The compiler's version of the shortcut spends 35 v_mov per line on pairing registers for the packed instructions
// and canonicalises every operand of fmaxf.  Here a line is 8 register PAIRS from start to end:
//   pre-scale   (x0, x1)            -> (x0, x1) / root                                   3 packed per pair
//   level 1     (x, y) of one pair  -> (x + y, x - y) -> / sqrt 2 = (sum, detail)        1 + 3 packed per pair
//   level 2..4  lows of two pairs   -> (a + b, a - b) -> / sqrt 2 = (sum, detail)        1 + 3 packed per butterfly
// with op_sel picking the halves, so nothing is ever moved: 84 packed instructions per line.  a - b is formed as
// a + (-b) (neg_hi), the same float.  gfx950 needs one wait state between a packed instruction and a consumer of its
// result (the compiler's own s_nop 0 in such chains): inside a block consumers are at least two instructions behind
// their producers except in the last butterfly, which carries explicit s_nop; every block ends with one.

// RN(1 / d) and -d in the low halves of scalar register pairs (the packed operand form op_sel_hi:[1,0] reads the low half twice)
struct DivConst {
    unsigned long long r, nd;
    __device__ __forceinline__ explicit DivConst(float d)
        : r(__builtin_amdgcn_readfirstlane(__float_as_uint(__fdiv_rn(1.0f, d)))),
          nd(__builtin_amdgcn_readfirstlane(__float_as_uint(-d))) {}
};

// x[i] / root for 8 pairs: q0 = x r; e = fma(-d, q0, x) (over x); q = fma(e, r, q0)
__device__ __forceinline__ void pk_prescale8(f2 (&x)[8], f2 (&q)[8], const DivConst& c) {
    asm("v_pk_mul_f32 %0, %8, %[r] op_sel_hi:[1,0]\n"
        "v_pk_mul_f32 %1, %9, %[r] op_sel_hi:[1,0]\n"
        "v_pk_mul_f32 %2, %10, %[r] op_sel_hi:[1,0]\n"
        "v_pk_mul_f32 %3, %11, %[r] op_sel_hi:[1,0]\n"
        "v_pk_mul_f32 %4, %12, %[r] op_sel_hi:[1,0]\n"
        "v_pk_mul_f32 %5, %13, %[r] op_sel_hi:[1,0]\n"
        "v_pk_mul_f32 %6, %14, %[r] op_sel_hi:[1,0]\n"
        "v_pk_mul_f32 %7, %15, %[r] op_sel_hi:[1,0]\n"
        "v_pk_fma_f32 %8, %0, %[nd], %8 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %9, %1, %[nd], %9 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %10, %2, %[nd], %10 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %11, %3, %[nd], %11 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %12, %4, %[nd], %12 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %13, %5, %[nd], %13 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %14, %6, %[nd], %14 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %15, %7, %[nd], %15 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %0, %8, %[r], %0 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %1, %9, %[r], %1 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %2, %10, %[r], %2 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %3, %11, %[r], %3 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %4, %12, %[r], %4 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %5, %13, %[r], %5 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %6, %14, %[r], %6 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %7, %15, %[r], %7 op_sel_hi:[1,0,1]\n"
        "s_nop 0"
        : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]), "=&v"(q[6]), "=&v"(q[7]),
          "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
        : [r] "s"(c.r), [nd] "s"(c.nd));
}

// level 1: every pair (x, y) -> ((x + y) / sqrt 2, (x - y) / sqrt 2); p is used up
__device__ __forceinline__ void pk_level1(f2 (&p)[8], f2 (&q)[8], const DivConst& c) {
    asm("v_pk_add_f32 %8, %8, %8 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]\n"
        "v_pk_add_f32 %9, %9, %9 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]\n"
        "v_pk_add_f32 %10, %10, %10 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]\n"
        "v_pk_add_f32 %11, %11, %11 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]\n"
        "v_pk_add_f32 %12, %12, %12 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]\n"
        "v_pk_add_f32 %13, %13, %13 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]\n"
        "v_pk_add_f32 %14, %14, %14 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]\n"
        "v_pk_add_f32 %15, %15, %15 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]\n"
        "v_pk_mul_f32 %0, %8, %[r] op_sel_hi:[1,0]\n"
        "v_pk_mul_f32 %1, %9, %[r] op_sel_hi:[1,0]\n"
        "v_pk_mul_f32 %2, %10, %[r] op_sel_hi:[1,0]\n"
        "v_pk_mul_f32 %3, %11, %[r] op_sel_hi:[1,0]\n"
        "v_pk_mul_f32 %4, %12, %[r] op_sel_hi:[1,0]\n"
        "v_pk_mul_f32 %5, %13, %[r] op_sel_hi:[1,0]\n"
        "v_pk_mul_f32 %6, %14, %[r] op_sel_hi:[1,0]\n"
        "v_pk_mul_f32 %7, %15, %[r] op_sel_hi:[1,0]\n"
        "v_pk_fma_f32 %8, %0, %[nd], %8 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %9, %1, %[nd], %9 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %10, %2, %[nd], %10 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %11, %3, %[nd], %11 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %12, %4, %[nd], %12 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %13, %5, %[nd], %13 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %14, %6, %[nd], %14 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %15, %7, %[nd], %15 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %0, %8, %[r], %0 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %1, %9, %[r], %1 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %2, %10, %[r], %2 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %3, %11, %[r], %3 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %4, %12, %[r], %4 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %5, %13, %[r], %5 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %6, %14, %[r], %6 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %7, %15, %[r], %7 op_sel_hi:[1,0,1]\n"
        "s_nop 0"
        : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]), "=&v"(q[6]), "=&v"(q[7]),
          "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7])
        : [r] "s"(c.r), [nd] "s"(c.nd));
}

// level 2: the sums (low halves) of pairs 2 i and 2 i + 1 -> ((a + b) / sqrt 2, (a - b) / sqrt 2), four times
__device__ __forceinline__ void pk_level2(const f2 (&p)[8], f2 (&q)[4], const DivConst& c) {
    f2 t0, t1, t2, t3;
    asm("v_pk_add_f32 %4, %8, %9 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]\n"
        "v_pk_add_f32 %5, %10, %11 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]\n"
        "v_pk_add_f32 %6, %12, %13 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]\n"
        "v_pk_add_f32 %7, %14, %15 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]\n"
        "v_pk_mul_f32 %0, %4, %[r] op_sel_hi:[1,0]\n"
        "v_pk_mul_f32 %1, %5, %[r] op_sel_hi:[1,0]\n"
        "v_pk_mul_f32 %2, %6, %[r] op_sel_hi:[1,0]\n"
        "v_pk_mul_f32 %3, %7, %[r] op_sel_hi:[1,0]\n"
        "v_pk_fma_f32 %4, %0, %[nd], %4 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %5, %1, %[nd], %5 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %6, %2, %[nd], %6 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %7, %3, %[nd], %7 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %0, %4, %[r], %0 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %1, %5, %[r], %1 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %2, %6, %[r], %2 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %3, %7, %[r], %3 op_sel_hi:[1,0,1]\n"
        "s_nop 0"
        : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]), [r] "s"(c.r), [nd] "s"(c.nd));
}

// levels 3 and 4; s3 and s4 return their dividends for the guard
__device__ __forceinline__ void pk_level34(const f2 (&p)[4], f2 (&q3)[2], f2& q4, f2 (&s3)[2], f2& s4, const DivConst& c) {
    f2 e0, e1, e4;
    asm("v_pk_add_f32 %3, %9, %10 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]\n"
        "v_pk_add_f32 %4, %11, %12 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]\n"
        "v_pk_mul_f32 %0, %3, %[r] op_sel_hi:[1,0]\n"
        "v_pk_mul_f32 %1, %4, %[r] op_sel_hi:[1,0]\n"
        "v_pk_fma_f32 %6, %0, %[nd], %3 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %7, %1, %[nd], %4 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %0, %6, %[r], %0 op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %1, %7, %[r], %1 op_sel_hi:[1,0,1]\n"
        "s_nop 0\n"
        "v_pk_add_f32 %5, %0, %1 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]\n"
        "s_nop 0\n"
        "v_pk_mul_f32 %2, %5, %[r] op_sel_hi:[1,0]\n"
        "s_nop 0\n"
        "v_pk_fma_f32 %8, %2, %[nd], %5 op_sel_hi:[1,0,1]\n"
        "s_nop 0\n"
        "v_pk_fma_f32 %2, %8, %[r], %2 op_sel_hi:[1,0,1]\n"
        "s_nop 0"
        : "=&v"(q3[0]), "=&v"(q3[1]), "=&v"(q4), "=&v"(s3[0]), "=&v"(s3[1]), "=&v"(s4), "=&v"(e0), "=&v"(e1), "=&v"(e4)
        : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), [r] "s"(c.r), [nd] "s"(c.nd));
}

// |a|, |b| into a running float maximum without the canonicalising v_max the compiler puts in front of fmaxf
__device__ __forceinline__ float max3_abs(float m, float a, float b) {
    float o;
    asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(o) : "v"(m), "v"(a), "v"(b));
    return o;
}

// One 16-value line: `load` fills the 16 inputs (it is called again when the line has to be redone).  `fast`
// (wave-uniform) tells the caller whether the shortcut held for the whole wave.
template <typename Load>
__device__ __forceinline__ float haar16(Load load, float (&d)[15], float root, float root2, bool& fast) {
    const DivConst c_root(root), c_root2(root2);
    {
        f2 x[8], q[8], u[8], v2[4], w[2], z, s3[2], s4;
        {
            float in[16];
            load(in);
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = f2{in[2 * i], in[2 * i + 1]};
        }
        LightGuard lg;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            lg.lo_in = min(lg.lo_in, min((__float_as_uint(x[i].x) << 1) - 1u, (__float_as_uint(x[i].y) << 1) - 1u));
            lg.hi = max3_abs(lg.hi, x[i].x, x[i].y);
        }
        pk_prescale8(x, q, c_root);
        pk_level1(q, u, c_root2);
        pk_level2(u, v2, c_root2);
        pk_level34(v2, w, z, s3, s4, c_root2);
        lg.levels34(s3[0], s3[1], s4);
        fast = !__any(lg.bad(z.x));
        if (fast) {
#pragma unroll
            for (int i = 0; i < 8; ++i) d[i] = u[i].y;
#pragma unroll
            for (int i = 0; i < 4; ++i) d[8 + i] = v2[i].y;
            d[12] = w[0].y; d[13] = w[1].y; d[14] = z.y;
            return z.x;
        }
    }
    // small inputs: the shortcut with every dividend checked; tiny / inf / NaN: true divisions.  The inputs are read
    // AGAIN (the barrier keeps the compiler from holding the first copies in registers across the light tier)
    asm volatile("" ::: "memory");
    float in[16];
    load(in);
    DivGuard g;
    float s4 = haar16_impl<true>(in, d, root, __fdiv_rn(1.0f, root), root2, __fdiv_rn(1.0f, root2), g);
    fast = !__any(g.bad());
    if (!fast) s4 = haar16_impl<false>(in, d, root, 0.0f, root2, 0.0f, g);
    return s4;
}

// The levels that pair the LANES lanes of one line (their level-4 averages).  At the level with lane distance m the
// lower lane of a pair continues with (lo + hi) / sqrt 2 and the upper one leaves with (lo - hi) / sqrt 2: ONE
// division per lane and level (until round 3 both lanes computed both).  lo - hi is formed as lo + (-hi), the same
// float for every input.  Returns the coefficient the lane ends up owning.
// the value of lane (id ^ M), M = 1, 2, 4: a DPP quad permutation or a ds_swizzle -- no address register, no bounds
// logic (what __shfl_xor compiles to: compare, select, shift, ds_bpermute)
template <int M>
__device__ __forceinline__ float lane_xor(float x) {
    static_assert(M == 1 || M == 2 || M == 4, "lane distance");
    const int i = __float_as_int(x);
    if constexpr (M == 1) return __int_as_float(__builtin_amdgcn_mov_dpp(i, 0xB1, 0xF, 0xF, true));         // quad_perm [1,0,3,2]
    else if constexpr (M == 2) return __int_as_float(__builtin_amdgcn_mov_dpp(i, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    else return __int_as_float(__builtin_amdgcn_ds_swizzle(i, 0x101F));                                      // bit mode: and 0x1F, xor 4
}

template <bool FAST, int LANES, int M = 1>
__device__ __forceinline__ void cross_level_step(float& cur, float& fin, bool& done, int h, float root2, float r_root2, DivGuard& g) {
    if constexpr (M < LANES) {
        const float other = lane_xor<M>(cur);
        const bool upper = (h & M) != 0;
        const float lo = upper ? other : cur, hi = upper ? -cur : other;
        const float s = __fadd_rn(lo, hi);
        if constexpr (FAST) g.dividends(s, s);
        const float q = div_c<FAST>(s, root2, r_root2);
        if (!done && upper) { fin = q; done = true; }
        cur = q;
        cross_level_step<FAST, LANES, 2 * M>(cur, fin, done, h, root2, r_root2, g);
    }
}

template <bool FAST, int LANES>
__device__ __forceinline__ float cross_levels_impl(float cur, int h, float root2, float r_root2, DivGuard& g) {
    float fin = 0.0f;
    bool done = false;
    cross_level_step<FAST, LANES>(cur, fin, done, h, root2, r_root2, g);
    if (!done) fin = cur;   // lane 0 keeps the line's average
    return fin;
}

// `fast`: the line's haar16 ran on the shortcut, i.e. its 16 inputs are finite and <= 2^126 -- the sums here stay
// finite (at most 2^127.5 after seven levels) and only the tiny-dividend guard is needed.
template <int LANES>
__device__ __forceinline__ float cross_levels(float cur, int h, float root2, bool fast) {
    if constexpr (LANES == 1) return cur;
    if (fast) {
        DivGuard g;
        const float fin = cross_levels_impl<true, LANES>(cur, h, root2, __fdiv_rn(1.0f, root2), g);
        if (!__any(g.bad())) return fin;
    }
    DivGuard g;
    return cross_levels_impl<false, LANES>(cur, h, root2, 0.0f, g);
}

// 32 columns: 72 VGPRs -> seven workgroups per CU, which is also what the 22 KB of LDS allow (90 VGPRs and five
// workgroups without the bound: 4.33 -> 3.81 ms per 500 k frames; the second bound is waves per SIMD).
// The sparse form (SPARSE, 32 bands): bands whose bin range is empty are +0.0 in every window (LBAudioDetective.m:386-404
// with an empty loop), and where only ONE of the left sixteen bands is live (44.1 kHz / 1024: band 13; bands 16, 18,
// 20..31 on the right) most of the row transform is structure, not arithmetic:
//   * row pass: ONE thread per row (two waves instead of four).  The right half is an ordinary 16-value line; the left
//     half is the single value walking up the levels -- at each level the pair (a, 0) or (0, a) gives the sum a / sqrt 2 and
//     the detail +-a / sqrt 2, every other detail of the half is (0 +- 0) / sqrt 2 = +0.0;
//   * the columns of the row transform's output that can be non-zero (21 of 32 there; the plan lists them) are the only
//     ones that go through the column transform, the search and the gather: the others' 128 coefficients are all +0.0,
//     and a zero never sets a Boolean (its sign code is 0) nor changes the rank of a non-zero coefficient (every non-zero
//     key is larger), whether it would have been selected or not -- frames with fewer than `keep` non-zero coefficients
//     included (tests/test_gpu_parity.py::test_stage2_corner_frames runs them through both forms);
//   * input: the compact frame stage 1 writes for such a plan (128 rows of the bands that can be non-zero: 15 of 32 at
//     44.1 kHz / 1024), less than half of the bytes;
//   * a workgroup is THREE waves (192 threads: 128 rows, up to 24 column slots of 8 threads) with 15 KB of transposed
//     coefficients instead of 20: nine workgroups fit a CU where the general form has seven.  That, not the smaller
//     instruction count, is what the sparse form returns: the kernel is bound by the life time of a workgroup (a chain
//     of fourteen barriers) times the number of workgroups a CU holds -- the first sparse build (256 threads, a wave
//     idle in every phase, 14 % fewer vector instructions) took exactly as long as the general form.
// Every value that is computed is computed by the same operations on the same operands as in the general form.
struct SparseArgs {
    uint32_t left, n_cols;
    uint32_t row_dw;               // bands a compact frame's row holds (Plan::Sparse::n_stored); a frame is 128 such rows
    uint32_t pos_left;             // position of the left half's live band in the row
    uint32_t pos_right[4];         // 16 bytes: position of band 16 + j in the row (0xFF: structurally empty, not stored)
    uint32_t cols[8];              // 32 ordered positions, one byte each: the columns that can be non-zero, ascending
    uint32_t rank[8];              // ordered position -> its place in that list (0xFF: structurally zero), one byte each
};
constexpr int kSparseCols = 21;    // column slots of the sparse form (every configuration with a compact layout has 21 live columns: 30 .. 60 kHz x
                                   // 512 / 1024 / 2048, tests/test_gpu_parity.py); a plan with more keeps the general form.  13.4 KB of
                                   // transposed coefficients and 64 VGPRs (4 of them spilled): TEN workgroups per CU (nine with 24 slots
                                   // and 66 VGPRs: 140 -> 137 us)
__device__ __forceinline__ uint32_t byte_of(const uint32_t (&tbl)[8], uint32_t i) { return (tbl[i >> 2] >> (8u * (i & 3u))) & 0xFFu; }

// RMASK (sparse form): the live bands of the right half as a compile-time mask (bit j: band 16 + j), with HAS_LEFT a live
// band on the left -- the row of a compact frame is then read with 16-byte loads and unpacked by name.  RMASK = 0: the
// positions come from the plan at run time (one 4-byte load per band: 146 us instead of 133 per 31 250 frames).
constexpr uint32_t kRmask44k = 0xFFF5u;   // 44.1 kHz / 1024: bands 16, 18, 20..31
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));

template <int COLS, bool SPARSE, uint32_t RMASK = 0u, bool HAS_LEFT = false>
__global__ __launch_bounds__(SPARSE ? 192 : COLS * 8, SPARSE ? 8 : COLS == 64 ? 6 : 7) void haar_select32_kernel(const float* __restrict__ frames, uint32_t keep,
                                                                          uint32_t subfp_len, uint32_t* __restrict__ packed,
                                                                          float* __restrict__ haar_out, const SparseArgs sp) {
    static_assert(!SPARSE || COLS == 32, "the sparse form is for 32 bands");
    constexpr int kCols = COLS;
    constexpr int kThreads = SPARSE ? 192 : COLS * 8;
    constexpr int kWavesPerWg = kThreads / 64;
    constexpr int H = COLS / 16;                     // threads per row in the row pass
    constexpr uint32_t kIdxBits = COLS == 16 ? 11 : COLS == 32 ? 12 : 13;   // flat index of a coefficient
    __shared__ __attribute__((aligned(16))) float s_t[(SPARSE ? kSparseCols : kCols) * 8 * kChunkDw];   // [col (sparse: slot)][chunk][20]
    __shared__ __attribute__((aligned(16))) unsigned long long s_cand[kCand];
    __shared__ uint32_t s_rank[kCand];
    __shared__ uint32_t s_red[16];
    __shared__ uint32_t s_ncand;
    __shared__ uint32_t s_bits[kPackedWords];
    __shared__ uint8_t s_pos[128];   // [i][j]: row position of coefficient i of the thread that owns chunk j (column pass)
    __shared__ uint32_t s_sel[4];    // the histogram's bracket: lo, hi, keys >= lo
    static_assert(sizeof(s_t) >= kHistBuckets * sizeof(uint32_t), "the histogram lives in the transposed coefficients' LDS");

    const int t = threadIdx.x;
    const uint64_t frame = blockIdx.x;
    const float root2 = __fsqrt_rn(2.0f);
#ifdef LBAD_EXP_TIMELINE
    long long ts[8];
    int n_steps = 0;
#define STAMP(i) ts[i] = __builtin_readcyclecounter()
#define COUNT_STEP() ++n_steps
#else
#define STAMP(i)
#define COUNT_STEP()
#endif
    STAMP(0);

    // the thread's sixteen frame values are asked for before anything else: the LDS set-up below runs while they travel.
    // Sparse form: the row of a compact frame holds only the bands that can be non-zero (sp.pos_right / sp.pos_left say
    // where; the plan's positions are uniform, so every load has a scalar offset or is skipped), the others are +0.0 here
    float pre[16];
    float pre_left = 0.0f;
    auto fetch_row = [&](float (&a)[16], float& left_value) {
        if constexpr (SPARSE && RMASK != 0u) {
            constexpr int kRight = __builtin_popcount(RMASK), kRow = kRight + (HAS_LEFT ? 1 : 0);
            const bool mine = t < (int)kRowsPerFrame;
            const float* rowp = frames + frame * (kRowsPerFrame * kRow) + (mine ? t : 0) * kRow;
            float got[kRow + 3];
#pragma unroll
            for (int q = 0; q + 4 <= kRow; q += 4) {
                const f4u v = *reinterpret_cast<const f4u*>(rowp + q);
                got[q] = v.x; got[q + 1] = v.y; got[q + 2] = v.z; got[q + 3] = v.w;
            }
            constexpr int kDone = kRow & ~3;
            if constexpr (kRow - kDone == 3) {
                const f3u v = *reinterpret_cast<const f3u*>(rowp + kDone);
                got[kDone] = v.x; got[kDone + 1] = v.y; got[kDone + 2] = v.z;
            } else if constexpr (kRow - kDone == 2) {
                const f2u v = *reinterpret_cast<const f2u*>(rowp + kDone);
                got[kDone] = v.x; got[kDone + 1] = v.y;
            } else if constexpr (kRow - kDone == 1) {
                got[kDone] = rowp[kDone];
            }
#pragma unroll
            for (int jj = 0; jj < 16; ++jj)
                a[jj] = ((RMASK >> jj) & 1u) ? got[__builtin_popcount(RMASK & ((1u << jj) - 1u))] : 0.0f;
            left_value = HAS_LEFT ? got[kRight] : 0.0f;
        } else if constexpr (SPARSE) {
            const bool mine = t < (int)kRowsPerFrame;
            const float* rowp = frames + frame * (kRowsPerFrame * sp.row_dw) + (mine ? t : 0) * sp.row_dw;
            // branch-free: an empty band reads the row's first float and drops it (sixteen loads in flight together)
            float got[16];
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const uint32_t pos = (sp.pos_right[jj >> 2] >> (8 * (jj & 3))) & 0xFFu;     // (uniform)
                got[jj] = rowp[pos != 0xFFu ? pos : 0u];
            }
            const float got_left = rowp[sp.left < 32u ? sp.pos_left : 0u];
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const uint32_t pos = (sp.pos_right[jj >> 2] >> (8 * (jj & 3))) & 0xFFu;
                a[jj] = pos != 0xFFu ? got[jj] : 0.0f;
            }
            left_value = sp.left < 32u ? got_left : 0.0f;
        } else {
            const float4* src = reinterpret_cast<const float4*>(frames + frame * (kRowsPerFrame * kCols) + (t / H) * kCols + 16 * (t % H));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = src[q];
                a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
            }
        }
    };
    fetch_row(pre, pre_left);
    bool first_read = true;

    if (t < (int)kPackedWords) s_bits[t] = 0;
    if (t < (int)kCand) s_rank[t] = 0;
    if (t == 0) s_ncand = 0;
    if (t < 128) {   // the closed form of pos[] below, as a table for step (b) of the gather
        const int i = t >> 3, jj = t & 7;
        const int cross = (jj & 1) ? 4 + (jj >> 1) : (jj & 2) ? 2 + (jj >> 2) : (jj ? 1 : 0);
        s_pos[t] = (uint8_t)(i < 8 ? 64 + 8 * jj + i : i < 12 ? 32 + 4 * jj + (i - 8) : i < 14 ? 16 + 2 * jj + (i - 12) : i == 14 ? 8 + jj : cross);
    }
    for (int i = t; i < 2 * (int)kCand; i += kThreads)
        reinterpret_cast<uint32_t*>(s_cand)[i] = 0;   // zero keys pad the list to a multiple of 8 for the ranking loop

    // ---- row pass, sparse form: thread = row (waves 0 and 1) ----------------------------------------------------
    if constexpr (SPARSE) {
        if (t < (int)kRowsPerFrame) {
            const int row = t;
            float d[15];
            auto load = [&](float (&a)[16]) {       // the first call takes what was fetched at the top, a redo reads again
                if (first_read) {
#pragma unroll
                    for (int jj = 0; jj < 16; ++jj) a[jj] = pre[jj];
                } else {
                    float again;
                    fetch_row(a, again);
                }
                first_read = false;
            };
            const float root = __fsqrt_rn(32.0f);
            bool fast;
            const float s4r = haar16(load, d, root, root2, fast);               // bands 16..31: levels 1..4
            // bands 0..15: the one live band's mean up the four levels.  At level k the pair is (a, 0) when bit k - 1 of
            // the band's index is clear -- sum (a + 0) / sqrt 2, detail (a - 0) / sqrt 2: the same quotient -- and (0, a)
            // when it is set: detail (0 - a) / sqrt 2.
            const float xl = pre_left;
            float det[4], s4l = 0.0f;
            auto chain = [&](auto fast_tag, DivGuard& g) {
                constexpr bool F = decltype(fast_tag)::value;
                const float r_root = F ? __fdiv_rn(1.0f, root) : 0.0f, r_root2 = F ? __fdiv_rn(1.0f, root2) : 0.0f;
                if constexpr (F) { g.inputs(xl, xl); g.dividends(xl, xl); }
                float a = div_c<F>(xl, root, r_root);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    // the pair is (a, 0) or (0, a): the sum is a + 0 either way, the difference a - 0 = a or 0 - a = -a, and a
                    // correctly rounded quotient of -x is minus the quotient of x -- ONE division serves both (the results
                    // differ from the two-division form in the sign of a zero at most, which nothing downstream can see:
                    // x +- (+-0) = x, and a zero coefficient has sign code 0 whatever its sign bit)
                    const bool odd = ((sp.left >> k) & 1u) != 0u;               // (uniform)
                    const float sum_in = __fadd_rn(a, 0.0f);
                    if constexpr (F) g.dividends(sum_in, sum_in);
                    const float sum = div_c<F>(sum_in, root2, r_root2);
                    det[k] = odd ? -sum : sum;
                    a = sum;
                }
                s4l = a;
            };
            // level 5 pairs the halves: the left one keeps (l + r) / sqrt 2, the right one leaves with (l - r) / sqrt 2 =
            // (l + (-r)) / sqrt 2 (cross_level_step's form)
            float avg = 0.0f, det5 = 0.0f;
            auto level5 = [&](auto fast_tag, DivGuard& g) {
                constexpr bool F = decltype(fast_tag)::value;
                const float r_root2 = F ? __fdiv_rn(1.0f, root2) : 0.0f;
                const float sp5 = __fadd_rn(s4l, s4r), sm5 = __fadd_rn(s4l, -s4r);
                if constexpr (F) g.dividends(sp5, sm5);
                avg = div_c<F>(sp5, root2, r_root2);
                det5 = div_c<F>(sm5, root2, r_root2);
            };
            {
                DivGuard g;
                chain(std::true_type{}, g);
                if (fast) level5(std::true_type{}, g);
                if (!fast || __any(g.bad())) {       // tiny / huge / NaN somewhere in the wave: true divisions
                    chain(std::false_type{}, g);
                    level5(std::false_type{}, g);
                }
            }
            // transposed store: coefficient at ordered position p of this row -> s_t[p][row >> 4][row & 15]; positions
            // that are structurally zero are neither written nor read
            float* base = s_t + (row >> 4) * kChunkDw + (row & 15);
            auto put = [&](uint32_t pos, float value) {                         // (pos and its slot are wave-uniform)
                const uint32_t slot = byte_of(sp.rank, pos);
                if (slot != 0xFFu) base[slot * (8 * kChunkDw)] = value;
            };
#pragma unroll
            for (int i = 0; i < 8; ++i) put(24 + i, d[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) put(12 + i, d[8 + i]);
#pragma unroll
            for (int i = 0; i < 2; ++i) put(6 + i, d[12 + i]);
            put(3, d[14]);
            if (sp.left < 32u) {
                put(16 + (sp.left >> 1), det[0]);
                put(8 + (sp.left >> 2), det[1]);
                put(4 + (sp.left >> 3), det[2]);
                put(2, det[3]);
            }
            put(0, avg);
            put(1, det5);
        }
    } else
    // ---- row pass: thread = (row, sixteenth h of the row) ------------------------------------------------
    {
        const int row = t / H, h = t % H;
        float d[15];
        auto load = [&](float (&a)[16]) {
            if (first_read) {
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) a[jj] = pre[jj];
            } else {
                float unused;
                fetch_row(a, unused);
            }
            first_read = false;
        };
        bool fast;
        const float cur = haar16(load, d, __fsqrt_rn((float)kCols), root2, fast);   // (16, 32, 64: the device root is exact for these)
        // the remaining levels pair the H sixteenths of the row; a lane leaves with its detail value as soon as
        // its index has the level's bit set, sixteenth 0 keeps the row's average
        const float fin = cross_levels<H>(cur, h, root2, fast);
        // sixteenth -> ordered position of its cross-lane value (H = 4: 0, 2, 1, 3)
        const int cross = H == 1 ? 0 : (h & 1) ? H / 2 + (h >> 1) : (h & 2) ? H / 4 + (h >> 2) : 0;
        // transposed store: coefficient at ordered position p of this row -> s_t[p][row >> 4][row & 15]
        float* base = s_t + (row >> 4) * kChunkDw + (row & 15);
#pragma unroll
        for (int i = 0; i < 8; ++i) base[(8 * H + 8 * h + i) * (8 * kChunkDw)] = d[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) base[(4 * H + 4 * h + i) * (8 * kChunkDw)] = d[8 + i];
#pragma unroll
        for (int i = 0; i < 2; ++i) base[(2 * H + 2 * h + i) * (8 * kChunkDw)] = d[12 + i];
        base[(H + h) * (8 * kChunkDw)] = d[14];
        base[cross * (8 * kChunkDw)] = fin;
    }
    __syncthreads();
    STAMP(1);

    // ---- column pass: thread = (column, chunk of 16 rows) -----------------------------------------
    // sparse form: the n_cols columns that can be non-zero, packed into the first 8 n_cols threads; the threads behind them
    // hold sixteen zeros (never selected: every threshold the search tries is > 0) and a wave without a column skips the
    // arithmetic altogether
    const bool live_thread = !SPARSE || (uint32_t)(t >> 3) < sp.n_cols;
    const bool live_wave = !SPARSE || (uint32_t)((t & ~63) >> 3) < sp.n_cols;          // (wave-uniform)
    const int cslot = SPARSE ? (live_thread ? t >> 3 : 0) : t >> 3;                   // where the column sits in s_t
    int col = t >> 3;                                                                 // its ordered position in a row
    if constexpr (SPARSE) col = (int)byte_of(sp.cols, (uint32_t)cslot);
    const int j = t & 7;
    float v[16];
    if (!live_wave) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = 0.0f;
    } else {
        float d[15];
        const float4* src = reinterpret_cast<const float4*>(s_t + (cslot * 8 + j) * kChunkDw);
        auto load = [&](float (&a)[16]) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 x = src[q];
                a[4 * q] = x.x; a[4 * q + 1] = x.y; a[4 * q + 2] = x.z; a[4 * q + 3] = x.w;
            }
        };
        bool fast;
        const float cur = haar16(load, d, __fsqrt_rn((float)kRowsPerFrame), root2, fast);
        // levels 5..7 across the 8 chunks of the column; chunk 0 keeps the overall average
        const float fin = cross_levels<8>(cur, j, root2, fast);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = d[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[8 + i] = d[8 + i];
        v[12] = d[12];
        v[13] = d[13];
        v[14] = d[14];
        v[15] = fin;
        if constexpr (SPARSE) {
            if (!live_thread) {
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = 0.0f;
            }
        }
    }
    // Ordered row position of the thread's i-th coefficient (levels 1..4 of its chunk, then the cross-lane value: chunk ->
    // 0,4,2,5,1,6,3,7).  Only the optional Haar tap and the plateau path need it; `jj` comes through an empty asm there so
    // that the compiler cannot compute the sixteen positions ahead of the branch for every frame.
    auto pos_of = [](int i, uint32_t jj) -> uint32_t {
        return i < 8 ? 64u + 8u * jj + (uint32_t)i : i < 12 ? 32u + 4u * jj + (uint32_t)(i - 8) : i < 14 ? 16u + 2u * jj + (uint32_t)(i - 12)
             : i == 14 ? 8u + jj : ((jj & 1u) ? 4u + (jj >> 1) : (jj & 2u) ? 2u + (jj >> 2) : (jj ? 1u : 0u));
    };
    auto opaque_j = [&]() -> uint32_t {
        uint32_t jj = (uint32_t)j;
        asm volatile("" : "+v"(jj));
        return jj;
    };
    if (haar_out && live_thread) {
        float* dst = haar_out + frame * (kRowsPerFrame * kCols);
        const uint32_t jj = opaque_j();
#pragma unroll
        for (int i = 0; i < 16; ++i) dst[pos_of(i, jj) * kCols + col] = v[i];
    }

    // key = the value's bits rotated left by one: |v| bits in the upper 31 bits, the sign in bit 0.  key >= 2 m  <=>
    // |v| bits >= m, so the search and the gather compare against doubled thresholds, and the coefficients themselves
    // need not stay in registers next to their keys.
    uint32_t key[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) key[i] = __builtin_rotateleft32(__float_as_uint(v[i]), 1);

    STAMP(2);
    // ---- threshold search ---------------------------------------------------------------------------
    uint32_t lo = 0, hi = 0x80000000u, cnt_lo = kRowsPerFrame * kCols;
    uint32_t idx_bound = kRowsPerFrame * kCols;
    int parity = 0;
#ifndef LBAD_EXP_NO_HISTOGRAM
    // First by HISTOGRAM (round 4): 2048 counters over the 11 leading bits of |v| (exponent + 3 mantissa bits: an eighth of
    // a binade each) in the LDS the transposed coefficients have just left; one wave sums them from the top and finds the
    // bucket b in which the count of keys >= its lower bound first reaches `keep` -- that bound is a threshold with
    // [keep, keep + population of b) keys above it, in one pass of 16 LDS atomics per thread instead of eleven rounds of 16
    // compares and a barrier (measured on synthetic and bird frames: 106-109 keys on average, never more than 126).  A
    // bucket that holds too many (a plateau, a dense cluster) leaves the bracket [b, b + 1) to the bisection below.  Bucket
    // 0 (zeros, denormals below 2^-126) is never counted: a threshold down there is the initial bracket's lower end.
    {
        uint32_t* hist = reinterpret_cast<uint32_t*>(s_t);
        __syncthreads();                                   // every wave has read (and, on the slow tiers, re-read) its lines
        for (int i = t; i < (int)(kHistBuckets / 4); i += kThreads) reinterpret_cast<uint4*>(hist)[i] = uint4{0u, 0u, 0u, 0u};
        __syncthreads();
        if (live_thread) {
            uint32_t any = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) any |= key[i];
            if (any > 1u) {                                // (a thread of zeros -- silence -- would hammer one counter)
#pragma unroll
                for (int i = 0; i < 16; ++i) atomicAdd(&hist[key[i] >> 21], 1u);
            }
        }
        __syncthreads();
        if (t < 64) {
            // lane l sums buckets 2047 - 32 l down to 2016 - 32 l; a prefix sum over the lanes is a suffix sum over buckets
            const uint4* src = reinterpret_cast<const uint4*>(hist) + (kHistBuckets / 4 - 8 - 8 * t);
            uint32_t own = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint4 c = src[q];
                own += (c.x + c.y) + (c.z + c.w);
            }
            if (t == 63) own -= hist[0];
            const uint32_t incl = wave_prefix_sum(own), excl = incl - own;
            const unsigned long long m1 = __ballot(excl < keep && keep <= incl);
            uint32_t out_lo = 0, out_hi = 1u << 20, out_cnt = kRowsPerFrame * kCols;
            if (m1 != 0ull) {                                                       // (wave-uniform)
                const int L = __ffsll((long long)m1) - 1;
                const uint32_t before = (uint32_t)__builtin_amdgcn_readlane((int)excl, L);
                const uint32_t top = kHistBuckets - 1u - 32u * (uint32_t)L;           // the lane's highest bucket
                const uint32_t h = t < 32 ? hist[top - (uint32_t)t] : 0u;
                const uint32_t p = before + wave_prefix_sum(h);
                const unsigned long long m2 = __ballot(t < 32 && p - h < keep && keep <= p);
                const int J = __ffsll((long long)m2) - 1;
                const uint32_t b = top - (uint32_t)J;
                if (m2 != 0ull && b != 0u) {
                    out_lo = b << 20;
                    out_hi = (b + 1u) << 20;
                    out_cnt = (uint32_t)__builtin_amdgcn_readlane((int)p, J);
                }
            }
            if (t == 0) { s_sel[0] = out_lo; s_sel[1] = out_hi; s_sel[2] = out_cnt; }
        }
        __syncthreads();
        lo = s_sel[0];
        hi = s_sel[1];
        cnt_lo = s_sel[2];
    }
#endif
    while (cnt_lo > kCand && hi - lo > 1) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        // one vector compare per key; the per-wave count is a scalar popcount of the lane mask, so the
        // adds and the cross-lane reduction run on the scalar unit instead of the VALU
        uint32_t c = 0;
        if (live_wave) {
#pragma unroll
            for (int i = 0; i < 16; ++i) c += (uint32_t)__popcll(__ballot(key[i] >= (mid << 1)));   // mid < 2^31
        }
        {
            uint32_t* slot = s_red + 8 * parity;
            if ((t & 63) == 0) slot[t >> 6] = c;
            __syncthreads();
            c = slots_sum<kWavesPerWg>(slot);
        }
        parity ^= 1;
        COUNT_STEP();
        if (c >= keep) { lo = mid; cnt_lo = c; } else { hi = mid; }
    }
    STAMP(3);
    // |v| bits above / equal to the threshold, in the keys' doubled domain (no shift of the keys: the compiler would
    // hoist sixteen of them in front of the rare paths that need them)
    const uint32_t lo2 = lo << 1;
    auto above = [&](uint32_t k) { return lo != 0x7FFFFFFFu && k >= lo2 + 2u; };
    auto tied = [&](uint32_t k) { return live_thread && k - lo2 < 2u; };      // (a thread without a column holds no coefficients)
    if (cnt_lo > kCand) {
        // plateau: more than kCand coefficients share the threshold key (e.g. digital silence).  Take
        // every key above it, then the tied ones in ascending flat-index order until `keep` is reached.
        uint32_t g = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) g += above(key[i]) ? 1u : 0u;
        g = block_sum<kWavesPerWg>(g, s_red, parity);
        parity ^= 1;
        const uint32_t jp = opaque_j();
        uint32_t ilo = 0, ihi = kRowsPerFrame * kCols;
        while (ilo < ihi) {
            const uint32_t im = ilo + ((ihi - ilo) >> 1);
            uint32_t c = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) c += (tied(key[i]) && pos_of(i, jp) * kCols + col < im) ? 1u : 0u;
            c = block_sum<kWavesPerWg>(c, s_red, parity);
            parity ^= 1;
            if (g + c >= keep) ihi = im; else ilo = im + 1;
        }
        idx_bound = ilo;
    }

    // ---- gather candidates, in two steps ---------------------------------------------------------------------
    // (a) every thread drops (key, where it came from) of its selected coefficients into the list: per coefficient a
    //     compare, the slot from lane-mask popcounts (one LDS atomic per wave) and one 8-byte store.  A wave holds ~26
    //     candidates among 1024 coefficients, so whatever else is done per coefficient here runs with one or two lanes
    //     active -- until round 3 that was the whole composite key (16 instructions, 16 times);
    // (b) one thread per CANDIDATE turns its entry into the composite
    //     |v| bits << (idx bits + 2) | (max idx - idx) << 2 | sign code        (sign code: 1 for v > 0, 2 for v < 0).
    {
        const uint32_t origin = ((uint32_t)col << 7) | (uint32_t)j;    // + (i << 3): index into s_pos, column above it
        typedef __attribute__((address_space(3))) void lds_void_t;
        uint32_t ncand_at = (uint32_t)(uintptr_t)(lds_void_t*)&s_ncand, one = 1u;
        asm volatile("" : "+v"(ncand_at), "+v"(one));                   // (held in registers, not rebuilt per block)
        auto gather = [&](auto selected) {
#ifdef LBAD_EXP_GATHER_BALLOT
            // the lane masks are taken once and kept (scalar registers) -- eight at a time, with one LDS atomic per wave and
            // half: sixteen masks at once push the kernel's arguments out of the scalar registers (v_writelane spills)
#pragma unroll
            for (int h0 = 0; h0 < 16; h0 += 8) {
                unsigned long long mask[8];
                uint32_t wave_total = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    mask[i] = __ballot(selected(h0 + i));
                    wave_total += (uint32_t)__popcll(mask[i]);
                }
                if (wave_total == 0) continue;                              // (wave-uniform)
                uint32_t base = 0;
                if ((t & 63) == 0) base = atomicAdd(&s_ncand, wave_total);
                base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const unsigned long long m = mask[i];
                    if (selected(h0 + i)) {
                        const uint32_t at = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                        s_cand[at] = ((unsigned long long)key[h0 + i] << 32) | (origin + (uint32_t)((h0 + i) << 3));
                    }
                    base += (uint32_t)__popcll(m);
                }
            }
#else
            // every selected coefficient takes its slot with ONE returning LDS atomic (inline asm: the compiler's atomic
            // optimiser would rebuild the ballot / mbcnt form around it, and it rematerialises the operands in every
            // block): the order of the list is whatever the hardware made it, the ranks below do not depend on it.  All
            // sixteen atomics are in flight before the first slot is used (one wait).
            uint32_t at[16];
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (selected(i)) asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(at[i]) : "v"(ncand_at), "v"(one) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(at[0]), "+v"(at[1]), "+v"(at[2]), "+v"(at[3]), "+v"(at[4]), "+v"(at[5]), "+v"(at[6]), "+v"(at[7]),
                           "+v"(at[8]), "+v"(at[9]), "+v"(at[10]), "+v"(at[11]), "+v"(at[12]), "+v"(at[13]), "+v"(at[14]), "+v"(at[15])
                         :: "memory");
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (selected(i)) s_cand[at[i]] = ((unsigned long long)key[i] << 32) | (origin + (uint32_t)(i << 3));
#endif
        };
        // (workgroup-uniform) without a plateau one compare per key decides; two copies of the loop, so that the masks
        // stay in scalar registers
        if (idx_bound == kRowsPerFrame * kCols) gather([&](int i) { return key[i] >= (lo << 1); });
        else {
            const uint32_t jg = opaque_j();
            gather([&](int i) { return above(key[i]) || (tied(key[i]) && pos_of(i, jg) * kCols + col < idx_bound); });
        }
    }
    __syncthreads();
    const uint32_t nc = s_ncand;
    if (t < (int)nc) {   // nc <= kCand <= kThreads
        const unsigned long long raw = s_cand[t];
        const uint32_t k = (uint32_t)(raw >> 32), from = (uint32_t)raw;
        const uint32_t idx = (uint32_t)s_pos[from & 127u] * kCols + (from >> 7);
        // |v| bits in [1, 0x7F800000] (not +-0, not NaN: neither v > 0 nor v < 0 holds for those)  <=>  k - 2 < 0xFF000000
        const uint32_t sg = k - 2u < 0xFF000000u ? (k & 1u) + 1u : 0u;
        s_cand[t] = ((unsigned long long)(k & ~1u) << (kIdxBits + 1)) | ((unsigned long long)(((1u << kIdxBits) - 1u) - idx) << 2) | sg;
    }
    __syncthreads();
    STAMP(4);

    // ---- rank: kThreads / 128 threads per candidate, each scans its share of the list --------------------
    {
        // 192 threads (sparse form): three parts of 64 threads, a thread ranks candidates i and i + 64 over its third of the
        // list -- both against the SAME loaded block of eight keys (round 4: one trip through LDS serves both)
        constexpr uint32_t kGroup = kThreads == 192 ? 64 : 128;                 // threads of a part
        constexpr uint32_t kParts = kThreads / kGroup;
        constexpr uint32_t kPer = kCand / kGroup;                               // candidates per thread: 2 or 1
        const uint32_t part = t / kGroup, i0 = t % kGroup;
        // blocks of 8 keys (the tail is zero-padded and never counts); each part takes its share of the blocks
        const uint32_t nblk = (nc + 7) >> 3, pblk = (nblk + kParts - 1) / kParts;
        const uint32_t b0 = part * pblk < nblk ? part * pblk : nblk, b1 = b0 + pblk < nblk ? b0 + pblk : nblk;
        unsigned long long mine[kPer];
        uint32_t r[kPer];
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) {
            const uint32_t i = i0 + k * kGroup;
            mine[k] = i < nc ? s_cand[i] : ~0ull;                               // (nothing is larger: rank 0, never stored)
            r[k] = 0;
        }
        for (uint32_t b = b0; b < b1; ++b) {
            const ulonglong2* src = reinterpret_cast<const ulonglong2*>(s_cand + 8 * b);
            const ulonglong2 c0 = src[0], c1 = src[1], c2 = src[2], c3 = src[3];
#pragma unroll
            for (uint32_t k = 0; k < kPer; ++k)
                r[k] += (c0.x > mine[k] ? 1u : 0u) + (c0.y > mine[k] ? 1u : 0u) + (c1.x > mine[k] ? 1u : 0u) + (c1.y > mine[k] ? 1u : 0u) +
                        (c2.x > mine[k] ? 1u : 0u) + (c2.y > mine[k] ? 1u : 0u) + (c3.x > mine[k] ? 1u : 0u) + (c3.y > mine[k] ? 1u : 0u);
        }
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k)
            if (r[k]) atomicAdd(&s_rank[i0 + k * kGroup], r[k]);
    }
    __syncthreads();
    if (t < (int)nc) {
        const uint32_t rank = s_rank[t];
        if (rank < keep) {
            const uint32_t sg = (uint32_t)s_cand[t] & 3u;
            const uint32_t b = 2 * rank;
            if (sg == 1u) atomicOr(&s_bits[b >> 5], 1u << (b & 31));
            else if (sg == 2u && b + 1 < subfp_len) atomicOr(&s_bits[(b + 1) >> 5], 1u << ((b + 1) & 31));
        }
    }
    __syncthreads();
    if (t < (int)kPackedWords) packed[frame * kPackedWords + t] = s_bits[t];
#ifdef LBAD_EXP_TIMELINE
    STAMP(5);
    if (haar_out && t == 0) {
        float* o = haar_out + frame * (kRowsPerFrame * kCols);
        for (int i = 1; i < 6; ++i) o[i - 1] = (float)(ts[i] - ts[i - 1]);
        o[5] = (float)n_steps;
        o[6] = (float)nc;
    }
#endif
}

}  // namespace

bool haar_select32_supported(const Plan& p) {
    return (p.bands == 16 || p.bands == 32 || p.bands == 64) && p.keep <= kCand && p.keep >= 1;
}

hipError_t launch_haar_select32(const Plan& plan, const float* d_frames, uint64_t n_frames, uint32_t* d_packed,
                                float* d_haar_out, hipStream_t stream, bool compact) {
    if (n_frames == 0) return hipSuccess;
    if (n_frames > 0x7fffffffull) return hipErrorInvalidValue;
    SparseArgs sp = {};
    if (compact) {
        if (!plan.sparse.ok || plan.bands != 32) return hipErrorInvalidValue;
        if (plan.sparse.n_cols > (uint32_t)kSparseCols) return hipErrorInvalidValue;
        sp.left = plan.sparse.left;
        sp.n_cols = plan.sparse.n_cols;
        sp.row_dw = plan.sparse.n_stored;
        sp.pos_left = plan.sparse.pos_left;
        for (uint32_t j = 0; j < 16; ++j) sp.pos_right[j >> 2] |= (uint32_t)plan.sparse.pos_right[j] << (8u * (j & 3u));
        for (uint32_t i = 0; i < 8; ++i) sp.rank[i] = 0xFFFFFFFFu;
        for (uint32_t c = 0; c < 32; ++c) sp.cols[c >> 2] |= (uint32_t)plan.sparse.cols[c] << (8u * (c & 3u));
        for (uint32_t c = 0; c < plan.sparse.n_cols; ++c) {
            const uint32_t pos = plan.sparse.cols[c];
            sp.rank[pos >> 2] = (sp.rank[pos >> 2] & ~(0xFFu << (8u * (pos & 3u)))) | (c << (8u * (pos & 3u)));
        }
        uint32_t rmask = 0;
        for (uint32_t j = 0; j < 16; ++j)
            if (plan.sparse.pos_right[j] != 0xFF) rmask |= 1u << j;
        if (rmask == kRmask44k && plan.sparse.left < 32)
            hipLaunchKernelGGL((haar_select32_kernel<32, true, kRmask44k, true>), dim3((uint32_t)n_frames), dim3(192), 0, stream, d_frames,
                               plan.keep, plan.subfp_len, d_packed, d_haar_out, sp);
        else
            hipLaunchKernelGGL((haar_select32_kernel<32, true>), dim3((uint32_t)n_frames), dim3(192), 0, stream, d_frames, plan.keep,
                               plan.subfp_len, d_packed, d_haar_out, sp);
        return hipGetLastError();
    }
    if (plan.bands == 16)
        hipLaunchKernelGGL((haar_select32_kernel<16, false>), dim3((uint32_t)n_frames), dim3(128), 0, stream, d_frames, plan.keep,
                           plan.subfp_len, d_packed, d_haar_out, sp);
    else if (plan.bands == 64)
        hipLaunchKernelGGL((haar_select32_kernel<64, false>), dim3((uint32_t)n_frames), dim3(512), 0, stream, d_frames, plan.keep,
                           plan.subfp_len, d_packed, d_haar_out, sp);
    else
        hipLaunchKernelGGL((haar_select32_kernel<32, false>), dim3((uint32_t)n_frames), dim3(256), 0, stream, d_frames, plan.keep,
                           plan.subfp_len, d_packed, d_haar_out, sp);
    return hipGetLastError();
}

}  // namespace lbad
