// fft64_lane.hpp -- the in-register part shared by the specialised stage-1 kernels: one lane holds 64
// complex points and runs DIT stages with compile-time W_64 twiddles (twiddle64.inc), in exactly the
// operation order of oracle/lbad_oracle.c:rfft_exec.
#pragma once

#include <hip/hip_runtime.h>

#include "twiddle64.inc"

namespace lbad {
namespace lane64 {

// A complex value is one even-aligned VGPR pair (.x = re, .y = im).  Butterflies are written on the
// pair so that they compile to v_pk_fma_f32 / v_pk_add_f32 with op_sel swizzles and SGPR twiddle
// pairs (two IEEE operations per instruction, no register shuffling); each half is the same
// correctly rounded fma / add the oracle performs.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef f32x2 cplx;

static __device__ __forceinline__ cplx fma2(cplx a, cplx b, cplx c) { return __builtin_elementwise_fma(a, b, c); }
static __device__ __forceinline__ cplx mk(float x, float y) {
    cplx r;
    r.x = x;
    r.y = y;
    return r;
}
typedef __attribute__((address_space(3))) volatile f32x2 lds_vf32x2;   // LDS, not mergeable into ds_*2_b64

static __device__ __forceinline__ constexpr int brev6(int v) {
    return ((v & 1) << 5) | ((v & 2) << 3) | ((v & 4) << 1) | ((v & 8) >> 1) | ((v & 16) >> 3) | ((v & 32) >> 5);
}

// full butterfly on registers, twiddle index t into the W_64 table (compile time)
template <int T>
static __device__ __forceinline__ void bfly(cplx& u, cplx& v) {
    const cplx a = u, b = v;
    if constexpr (T == 0) {
        u = a + b;
        v = a - b;
    } else if constexpr (T == 16) {  // w = -i: w b = (b.y, -b.x); 1 * x + y rounds exactly like x + y
        u = fma2(mk(1.0f, -1.0f), b.yx, a);
        v = fma2(mk(-1.0f, 1.0f), b.yx, a);
    } else {
        constexpr float wr = kTw64Re[T], wi = kTw64Im[T];
        const cplx bs = b.yx;
        // u.x = fma(wr, b.x, fma(-wi, b.y, a.x)), u.y = fma(wr, b.y, fma(wi, b.x, a.y)); v likewise negated
        u = fma2(mk(wr, wr), b, fma2(mk(-wi, wi), bs, a));
        v = fma2(mk(-wr, -wr), b, fma2(mk(wi, -wi), bs, a));
    }
}

template <int S, int BASE, int J>
static __device__ __forceinline__ void stage_block_j(cplx (&x)[64]) {
    constexpr int half = 1 << (S - 1);
    if constexpr (J < half) {
        bfly<J*(64 >> S)>(x[BASE + J], x[BASE + J + half]);
        stage_block_j<S, BASE, J + 1>(x);
    }
}

template <int S, int BASE>
static __device__ __forceinline__ void stage_blocks(cplx (&x)[64]) {
    if constexpr (BASE < 64) {
        stage_block_j<S, BASE, 0>(x);
        stage_blocks<S, BASE + (1 << S)>(x);
    }
}

// u + w v with a run-time twiddle (tree stages use per-lane tables)
static __device__ __forceinline__ cplx madd(cplx u, float wr, float wi, cplx v) {
    return fma2(mk(wr, wr), v, fma2(mk(-wi, wi), v.yx, u));
}

// u + w v and u - w v with the twiddle as the (wr, wi) pair it is read as (round 5): op_sel broadcasts wi, then wr, to both
// halves, op_sel on v swaps (v.y, v.x), neg_lo / neg_hi carry the signs -- the nesting and the operands of madd() / msub(),
// i.e. fma(wr, v, fma(-+wi, v.yx, u)) per half, without the two or three moves per twiddle the compiler spends on
// building (-wi, wi) and (wi, -wi) in registers.
// (the inner fmas of both results first: a packed fma that follows the one it depends on directly costs a wait state)
static __device__ __forceinline__ void bfly_w(cplx u, cplx w, cplx v, cplx& plus, cplx& minus) {
    cplx tp, tm;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(tp) : "v"(w), "v"(v), "v"(u));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "=v"(tm) : "v"(w), "v"(v), "v"(u));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(plus) : "v"(w), "v"(v), "v"(tp));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(minus) : "v"(w), "v"(v), "v"(tm));
}

// the two results on their own (pruned stages need only one of them)
static __device__ __forceinline__ cplx madd_w(cplx u, cplx w, cplx v) {
    cplx t, r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(t) : "v"(w), "v"(v), "v"(u));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(w), "v"(v), "v"(t));
    return r;
}
static __device__ __forceinline__ cplx msub_w(cplx u, cplx w, cplx v) {
    cplx t, r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "=v"(t) : "v"(w), "v"(v), "v"(u));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "v"(w), "v"(v), "v"(t));
    return r;
}

}  // namespace lane64
}  // namespace lbad
