// stream_common.hpp -- pieces shared by the streaming stage-1 kernels (k_rows_stream.hip: 4096-sample windows,
// k_rows_stream2.hip: 2048-sample windows): a wave walks the windows of a clip in time and keeps the
// sub-transforms consecutive windows share.
#pragma once

#include "internal.hpp"
#include "fft64_lane.hpp"

namespace lbad {
namespace stream {

using namespace lane64;

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

// 16 points idx + STEP m', m' = 0..15, into the bit-reversed slots of a 16-point DIT
template <int FMT, int STEP, int T>
__device__ __forceinline__ void load16(cplx (&x)[16], const void* p, int64_t idx) {
    if constexpr (T < 16) {
#ifdef LBAD_EXP_NOLOADS
        x[T] = mk((float)(idx & 1023) * 1e-3f, (float)T);
#elif defined(LBAD_EXP_HOTLOADS)
        x[T] = load_point<FMT>(p, (idx & 2047) + STEP * brev4(T));      // every wave re-reads the same 32 KB: L2 / L1 hits
#else
        x[T] = load_point<FMT>(p, idx + STEP * brev4(T));    // slot T holds point m' = brev4(T)
#endif
        load16<FMT, STEP, T + 1>(x, p, idx);
    }
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

// sm = a + conj(b), df = a - conj(b): one packed add each with the sign on one half of b (the compiler builds
// conj(b) in registers first when this is written on the vector type)
__device__ __forceinline__ void add_conj(const cplx a, const cplx b, cplx& sm, cplx& df) {
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(sm) : "v"(a), "v"(b));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(df) : "v"(a), "v"(b));
}

__device__ __forceinline__ cplx msub(cplx u, float wr, float wi, cplx v) {     // u - w v
    return fma2(mk(-wr, -wr), v, fma2(mk(wi, -wi), v.yx, u));
}

}  // namespace stream
}  // namespace lbad
