// Does the 2-cycle issue of a wave64 VALU instruction (tools/ubench/valu_rates.hip: v_bitop3 / v_and / v_add_f32 / v_fma_f32 ...)
// survive THREE DISTINCT vector sources?  Round 6: compare_short_multi_kernel was planned on 2 cycles per v_bitop3 and ran at
// 4.  Explicit registers: sources v[0..11] (never written), destinations v[16..23].
// build: hipcc --offload-arch=gfx950 -O3 -o operand_rates operand_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23"

// eight instructions, destinations v16..v23
#define B3(d, a, b, c) "v_bitop3_b32 v" #d ", v" #a ", v" #b ", v" #c " bitop3:0x96\n"
#define FMA(d, a, b, c) "v_fma_f32 v" #d ", v" #a ", v" #b ", v" #c "\n"
#define AND2(d, a, b) "v_and_b32 v" #d ", v" #a ", v" #b "\n"
#define ADDF(d, a, b) "v_add_f32 v" #d ", v" #a ", v" #b "\n"
#define PKF(d, a, b, c) "v_pk_fma_f32 v[" #d ":" #d "+1], v[" #a ":" #a "+1], v[" #b ":" #b "+1], v[" #c ":" #c "+1]\n"
#define PKA(d, a, b) "v_pk_add_f32 v[" #d ":" #d "+1], v[" #a ":" #a "+1], v[" #b ":" #b "+1]\n"
#define PKFS(d, a, c) "v_pk_fma_f32 v[" #d ":" #d "+1], v[" #a ":" #a "+1], s[4:5], v[" #c ":" #c "+1]\n"
#define BCNT(d, a, b) "v_bcnt_u32_b32 v" #d ", v" #a ", v" #b "\n"

static __device__ __forceinline__ void init_regs() {
    asm volatile("v_mov_b32 v0, 1\nv_mov_b32 v1, 2\nv_mov_b32 v2, 3\nv_mov_b32 v3, 4\nv_mov_b32 v4, 5\nv_mov_b32 v5, 6\n"
                 "v_mov_b32 v6, 7\nv_mov_b32 v7, 8\nv_mov_b32 v8, 9\nv_mov_b32 v9, 10\nv_mov_b32 v10, 11\nv_mov_b32 v11, 12\n"
                 "v_mov_b32 v16, 0\nv_mov_b32 v17, 0\nv_mov_b32 v18, 0\nv_mov_b32 v19, 0\nv_mov_b32 v20, 0\nv_mov_b32 v21, 0\n"
                 "v_mov_b32 v22, 0\nv_mov_b32 v23, 0\n" ::: CLOB);
}

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters) {
    init_regs();
    for (int it = 0; it < iters; ++it) {
        if (OP == 0)        // three distinct sources, banks 0 1 2 (register number mod 4)
            { REP8(asm volatile(B3(16, 0, 1, 2) B3(17, 0, 1, 2) B3(18, 0, 1, 2) B3(19, 0, 1, 2) B3(20, 0, 1, 2) B3(21, 0, 1, 2) B3(22, 0, 1, 2) B3(23, 0, 1, 2) ::: CLOB);) }
        else if (OP == 1)   // three distinct sources, one bank
            { REP8(asm volatile(B3(16, 0, 4, 8) B3(17, 0, 4, 8) B3(18, 0, 4, 8) B3(19, 0, 4, 8) B3(20, 0, 4, 8) B3(21, 0, 4, 8) B3(22, 0, 4, 8) B3(23, 0, 4, 8) ::: CLOB);) }
        else if (OP == 2)   // two distinct sources (a, b, a)
            { REP8(asm volatile(B3(16, 0, 1, 0) B3(17, 0, 1, 0) B3(18, 0, 1, 0) B3(19, 0, 1, 0) B3(20, 0, 1, 0) B3(21, 0, 1, 0) B3(22, 0, 1, 0) B3(23, 0, 1, 0) ::: CLOB);) }
        else if (OP == 3)   // destination is a source (d, a, d) -- the form valu_rates.hip measured
            { REP8(asm volatile(B3(16, 16, 1, 16) B3(17, 17, 1, 17) B3(18, 18, 1, 18) B3(19, 19, 1, 19) B3(20, 20, 1, 20) B3(21, 21, 1, 21) B3(22, 22, 1, 22) B3(23, 23, 1, 23) ::: CLOB);) }
        else if (OP == 4)   // three distinct sources that change from instruction to instruction (the scan's shape)
            { REP8(asm volatile(B3(16, 0, 1, 2) B3(17, 3, 4, 5) B3(18, 6, 7, 8) B3(19, 9, 10, 11) B3(20, 1, 2, 3) B3(21, 4, 5, 6) B3(22, 7, 8, 9) B3(23, 10, 11, 0) ::: CLOB);) }
        else if (OP == 5)   // v_fma_f32, three distinct sources
            { REP8(asm volatile(FMA(16, 0, 1, 2) FMA(17, 0, 1, 2) FMA(18, 0, 1, 2) FMA(19, 0, 1, 2) FMA(20, 0, 1, 2) FMA(21, 0, 1, 2) FMA(22, 0, 1, 2) FMA(23, 0, 1, 2) ::: CLOB);) }
        else if (OP == 6)   // v_and_b32, two distinct sources, destination apart
            { REP8(asm volatile(AND2(16, 0, 1) AND2(17, 0, 1) AND2(18, 0, 1) AND2(19, 0, 1) AND2(20, 0, 1) AND2(21, 0, 1) AND2(22, 0, 1) AND2(23, 0, 1) ::: CLOB);) }
        else if (OP == 7)   // v_add_f32, two distinct sources, destination apart
            { REP8(asm volatile(ADDF(16, 0, 1) ADDF(17, 0, 1) ADDF(18, 0, 1) ADDF(19, 0, 1) ADDF(20, 0, 1) ADDF(21, 0, 1) ADDF(22, 0, 1) ADDF(23, 0, 1) ::: CLOB);) }
        else if (OP == 8)   // the scan's pair: two v_bitop3 (second reads the first's result) + accumulating v_bcnt
            { REP8(asm volatile(B3(16, 0, 1, 2) B3(17, 3, 4, 5) B3(16, 16, 1, 6) B3(17, 17, 4, 7) BCNT(20, 16, 20) BCNT(21, 17, 21)
                              B3(18, 8, 9, 2) B3(18, 18, 9, 6) BCNT(22, 18, 22) ::: CLOB);) }
        else if (OP == 9)   // two sources equal (a, a, b): does a shared register count once?
            { REP8(asm volatile(B3(16, 0, 0, 1) B3(17, 0, 0, 1) B3(18, 0, 0, 1) B3(19, 0, 0, 1) B3(20, 0, 0, 1) B3(21, 0, 0, 1) B3(22, 0, 0, 1) B3(23, 0, 0, 1) ::: CLOB);) }

        else if (OP == 10)  // sources on banks 0 0 1 (first two share a bank)
            { REP8(asm volatile(B3(16, 0, 4, 1) B3(17, 0, 4, 1) B3(18, 0, 4, 1) B3(19, 0, 4, 1) B3(20, 0, 4, 1) B3(21, 0, 4, 1) B3(22, 0, 4, 1) B3(23, 0, 4, 1) ::: CLOB);) }
        else if (OP == 11)  // banks 0 1 0 (first and third)
            { REP8(asm volatile(B3(16, 0, 1, 4) B3(17, 0, 1, 4) B3(18, 0, 1, 4) B3(19, 0, 1, 4) B3(20, 0, 1, 4) B3(21, 0, 1, 4) B3(22, 0, 1, 4) B3(23, 0, 1, 4) ::: CLOB);) }
        else if (OP == 12)  // banks 1 0 0 (second and third)
            { REP8(asm volatile(B3(16, 1, 0, 4) B3(17, 1, 0, 4) B3(18, 1, 0, 4) B3(19, 1, 0, 4) B3(20, 1, 0, 4) B3(21, 1, 0, 4) B3(22, 1, 0, 4) B3(23, 1, 0, 4) ::: CLOB);) }
        else if (OP == 13)  // v_fma_f32, one bank
            { REP8(asm volatile(FMA(16, 0, 4, 8) FMA(17, 0, 4, 8) FMA(18, 0, 4, 8) FMA(19, 0, 4, 8) FMA(20, 0, 4, 8) FMA(21, 0, 4, 8) FMA(22, 0, 4, 8) FMA(23, 0, 4, 8) ::: CLOB);) }
        else if (OP == 14)  // v_and_b32, both sources on one bank
            { REP8(asm volatile(AND2(16, 0, 4) AND2(17, 0, 4) AND2(18, 0, 4) AND2(19, 0, 4) AND2(20, 0, 4) AND2(21, 0, 4) AND2(22, 0, 4) AND2(23, 0, 4) ::: CLOB);) }
        else if (OP == 15)  // v_bcnt, sources on different banks, destination apart
            { REP8(asm volatile(BCNT(16, 0, 1) BCNT(17, 0, 1) BCNT(18, 0, 1) BCNT(19, 0, 1) BCNT(20, 0, 1) BCNT(21, 0, 1) BCNT(22, 0, 1) BCNT(23, 0, 1) ::: CLOB);) }
        else if (OP == 16)  // the scan's pair with every source triple on three banks: 8 bitop3 + 4 bcnt
            { REP8(asm volatile(B3(16, 0, 2, 1) B3(17, 16, 2, 3) BCNT(20, 17, 20) B3(18, 4, 6, 5) B3(19, 18, 6, 7) BCNT(20, 19, 20)
                              B3(16, 8, 10, 9) B3(17, 16, 10, 11) BCNT(21, 17, 21) B3(18, 0, 6, 9) B3(19, 18, 6, 3) BCNT(21, 19, 21) ::: CLOB);) }

        else if (OP == 17)  // grouped: 8 bitop3 (conflict-free), then 4 bcnt
            { REP8(asm volatile(B3(16, 0, 2, 1) B3(18, 4, 6, 5) B3(12, 8, 10, 9) B3(14, 0, 6, 9) B3(17, 16, 2, 3) B3(19, 18, 6, 7) B3(13, 12, 10, 11) B3(15, 14, 6, 3)
                              BCNT(20, 17, 20) BCNT(21, 19, 21) BCNT(22, 13, 22) BCNT(23, 15, 23) ::: CLOB);) }
        else if (OP == 18)  // grouped x 2: 16 bitop3, then 8 bcnt
            { REP8(asm volatile(B3(16, 0, 2, 1) B3(18, 4, 6, 5) B3(12, 8, 10, 9) B3(14, 0, 6, 9) B3(17, 16, 2, 3) B3(19, 18, 6, 7) B3(13, 12, 10, 11) B3(15, 14, 6, 3)
                              B3(16, 0, 2, 1) B3(18, 4, 6, 5) B3(12, 8, 10, 9) B3(14, 0, 6, 9) B3(16, 16, 2, 3) B3(18, 18, 6, 7) B3(12, 12, 10, 11) B3(14, 14, 6, 3)
                              BCNT(20, 17, 20) BCNT(21, 19, 21) BCNT(22, 13, 22) BCNT(23, 15, 23) BCNT(20, 16, 20) BCNT(21, 18, 21) BCNT(22, 12, 22) BCNT(23, 14, 23) ::: CLOB);) }
        else if (OP == 19)  // strictly alternating fast / slow: bitop3, bcnt, bitop3, bcnt ...
            { REP8(asm volatile(B3(16, 0, 2, 1) BCNT(20, 17, 20) B3(18, 4, 6, 5) BCNT(21, 19, 21) B3(12, 8, 10, 9) BCNT(22, 13, 22) B3(14, 0, 6, 9) BCNT(23, 15, 23) ::: CLOB);) }
        else if (OP == 20)  // fast only, but DEPENDENT pairs back to back (second reads the first's result)
            { REP8(asm volatile(B3(16, 0, 2, 1) B3(17, 16, 2, 3) B3(18, 4, 6, 5) B3(19, 18, 6, 7) B3(12, 8, 10, 9) B3(13, 12, 10, 11) B3(14, 0, 6, 9) B3(15, 14, 6, 3) ::: CLOB);) }

        else if (OP == 21)  // v_pk_fma_f32, the three source pairs on banks (0 1) (2 3) (0 1)
            { REP8(asm volatile(PKF(16, 0, 2, 4) PKF(18, 0, 2, 4) PKF(20, 0, 2, 4) PKF(22, 0, 2, 4) PKF(16, 0, 2, 4) PKF(18, 0, 2, 4) PKF(20, 0, 2, 4) PKF(22, 0, 2, 4) ::: CLOB);) }
        else if (OP == 22)  // v_pk_fma_f32, all three source pairs on banks (0 1)
            { REP8(asm volatile(PKF(16, 0, 4, 8) PKF(18, 0, 4, 8) PKF(20, 0, 4, 8) PKF(22, 0, 4, 8) PKF(16, 0, 4, 8) PKF(18, 0, 4, 8) PKF(20, 0, 4, 8) PKF(22, 0, 4, 8) ::: CLOB);) }
        else if (OP == 23)  // v_pk_fma_f32, two sources the same pair
            { REP8(asm volatile(PKF(16, 0, 0, 2) PKF(18, 0, 0, 2) PKF(20, 0, 0, 2) PKF(22, 0, 0, 2) PKF(16, 0, 0, 2) PKF(18, 0, 0, 2) PKF(20, 0, 0, 2) PKF(22, 0, 0, 2) ::: CLOB);) }
        else if (OP == 24)  // v_pk_add_f32, source pairs on banks (0 1) (2 3)
            { REP8(asm volatile(PKA(16, 0, 2) PKA(18, 0, 2) PKA(20, 0, 2) PKA(22, 0, 2) PKA(16, 0, 2) PKA(18, 0, 2) PKA(20, 0, 2) PKA(22, 0, 2) ::: CLOB);) }
        else if (OP == 25)  // v_pk_add_f32, both source pairs on banks (0 1)
            { REP8(asm volatile(PKA(16, 0, 4) PKA(18, 0, 4) PKA(20, 0, 4) PKA(22, 0, 4) PKA(16, 0, 4) PKA(18, 0, 4) PKA(20, 0, 4) PKA(22, 0, 4) ::: CLOB);) }
        else if (OP == 26)  // v_pk_fma_f32 with an SGPR pair as the middle source
            { REP8(asm volatile(PKFS(16, 0, 4) PKFS(18, 0, 4) PKFS(20, 0, 4) PKFS(22, 0, 4) PKFS(16, 0, 4) PKFS(18, 0, 4) PKFS(20, 0, 4) PKFS(22, 0, 4) ::: CLOB);) }
    }
    unsigned s;
    asm volatile("v_add_u32 %0, v16, v17\nv_add_u32 %0, %0, v18\nv_add_u32 %0, %0, v19\nv_add_u32 %0, %0, v20\nv_add_u32 %0, %0, v21\n"
                 "v_add_u32 %0, %0, v22\nv_add_u32 %0, %0, v23\n" : "=v"(s) :: CLOB);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
static const char* kNames[] = {"bitop3 3 distinct, banks 0 1 2", "bitop3 3 distinct, one bank", "bitop3 a b a", "bitop3 d a d (dest = source)",
                               "bitop3 3 distinct, varying", "v_fma_f32 3 distinct", "v_and_b32 2 distinct", "v_add_f32 2 distinct",
                               "scan pair mix (6 bitop3 + 3 bcnt)", "bitop3 a a b", "bitop3 banks 0 0 1", "bitop3 banks 0 1 0", "bitop3 banks 1 0 0", "v_fma_f32 one bank", "v_and_b32 one bank", "v_bcnt two banks", "pair mix, conflict-free (8 bitop3 + 4 bcnt)", "grouped 8 bitop3 then 4 bcnt", "grouped 16 bitop3 then 8 bcnt", "alternating bitop3 / bcnt (4 + 4)", "8 bitop3, dependent pairs", "v_pk_fma_f32 banks (01)(23)(01)", "v_pk_fma_f32 all pairs on (01)", "v_pk_fma_f32 a a b", "v_pk_add_f32 banks (01)(23)", "v_pk_add_f32 both on (01)", "v_pk_fma_f32 v s v"};
static const int kPer[] = {8, 8, 8, 8, 8, 8, 8, 8, 9, 8, 8, 8, 8, 8, 8, 8, 12, 12, 24, 8, 8, 8, 8, 8, 8, 8, 8};
template <int OP>
void run(unsigned* d_out) {
    const int iters = 1000;
    double ns[2];
    int w = 0;
    for (int waves = 4; waves <= 8; waves *= 2, ++w) {
        hipLaunchKernelGGL(k<OP>, dim3(256 * waves), dim3(256), 0, 0, d_out, iters);
        (void)hipDeviceSynchronize();
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(256 * waves), dim3(256), 0, 0, d_out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        ns[w] = ms * 1e6 / ((double)iters * 8 * kPer[OP] * waves);
    }
    printf("%-36s %5.2f / %5.2f ns per instruction and SIMD at 4 / 8 waves per SIMD\n", kNames[OP], ns[0], ns[1]);
    if constexpr (OP + 1 < 27) run<OP + 1>(d_out);
}
int main() {
    unsigned* d_out;
    (void)hipMalloc(&d_out, 256 * 8 * 256 * 4);
    run<0>(d_out);
    return 0;
}
