// Issue rate of the vector instructions the integer-heavy kernels are made of (k_haar_select32.hip's search / gather /
// guards, k_sliding.hip's compare chain): 8 independent chains per wave, 4 and 8 waves per SIMD.  The guide's "a wave64
// VALU instruction issues over 2 cycles" holds for some opcodes only; the others take 4.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

// operands: %0..%7 = f[0..7] (32-bit), %8..%15 = r[0..7] (64-bit), %16 = c (32-bit), %17 = c64
#define OPS8(A) A(0, 8) A(1, 9) A(2, 10) A(3, 11) A(4, 12) A(5, 13) A(6, 14) A(7, 15)
#define REP8(x) x x x x x x x x
#define BODY(A)                                                                                                         \
    for (int it = 0; it < iters; ++it) {                                                                                \
        REP8(asm volatile(OPS8(A)                                                                                       \
                          : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), \
                            "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])  \
                          : "v"(c), "v"(c64)                                                                            \
                          : "vcc");)                                                                                    \
    }

// one macro per instruction: d = 32-bit register, q = 64-bit register
#define I_AND(d, q) "v_and_b32 %" #d ", %" #d ", %16\n"
#define I_XOR(d, q) "v_xor_b32 %" #d ", %" #d ", %16\n"
#define I_ADD(d, q) "v_add_u32 %" #d ", %" #d ", %16\n"
#define I_SHL(d, q) "v_lshlrev_b32 %" #d ", 1, %" #d "\n"
#define I_MIN(d, q) "v_min_u32 %" #d ", %" #d ", %16\n"
#define I_MIN3(d, q) "v_min3_u32 %" #d ", %" #d ", %16, %" #d "\n"
#define I_LSHLADD(d, q) "v_lshl_add_u32 %" #d ", %" #d ", 1, %16\n"
#define I_LSHLOR(d, q) "v_lshl_or_b32 %" #d ", %" #d ", 1, %16\n"
#define I_ANDOR(d, q) "v_and_or_b32 %" #d ", %" #d ", %16, %" #d "\n"
#define I_ADD3(d, q) "v_add3_u32 %" #d ", %" #d ", %16, %" #d "\n"
#define I_BFE(d, q) "v_bfe_u32 %" #d ", %" #d ", 1, 5\n"
#define I_ALIGN(d, q) "v_alignbit_b32 %" #d ", %" #d ", %" #d ", 31\n"
#define I_BITOP3(d, q) "v_bitop3_b32 %" #d ", %" #d ", %16, %" #d " bitop3:0x96\n"
#define I_BCNT(d, q) "v_bcnt_u32_b32 %" #d ", %16, %" #d "\n"
#define I_MBCNT(d, q) "v_mbcnt_lo_u32_b32 %" #d ", %16, %" #d "\n"
#define I_CNDMASK(d, q) "v_cndmask_b32 %" #d ", %" #d ", %16, vcc\n"
#define I_CMP(d, q) "v_cmp_ge_u32 vcc, %" #d ", %16\n"
#define I_CMP64(d, q) "v_cmp_gt_u64 vcc, %" #q ", %17\n"
#define I_ADDC(d, q) "v_addc_co_u32 %" #d ", vcc, %" #d ", %16, vcc\n"
#define I_MAX3F(d, q) "v_max3_f32 %" #d ", %" #d ", |%16|, |%" #d "|\n"
#define I_MAXF(d, q) "v_max_f32 %" #d ", %" #d ", %16\n"
#define I_ADDF(d, q) "v_add_f32 %" #d ", %" #d ", %16\n"
#define I_ADDF_ABS(d, q) "v_add_f32 %" #d ", |%" #d "|, -|%16|\n"
#define I_MULF(d, q) "v_mul_f32 %" #d ", %" #d ", %16\n"
#define I_FMAF(d, q) "v_fma_f32 %" #d ", %" #d ", %16, %" #d "\n"
#define I_FMACF(d, q) "v_fmac_f32 %" #d ", %16, %16\n"
#define I_PKADD(d, q) "v_pk_add_f32 %" #q ", %" #q ", %17\n"
#define I_PKMUL(d, q) "v_pk_mul_f32 %" #q ", %" #q ", %17\n"
#define I_PKFMA(d, q) "v_pk_fma_f32 %" #q ", %" #q ", %17, %" #q "\n"
#define I_PERM(d, q) "v_perm_b32 %" #d ", %" #d ", %16, %" #d "\n"
#define I_MAD24(d, q) "v_mad_u32_u24 %" #d ", %" #d ", %16, %" #d "\n"
#define I_MULLO(d, q) "v_mul_lo_u32 %" #d ", %" #d ", %16\n"
#define I_MOV(d, q) "v_mov_b32 %" #d ", %16\n"
#define I_CVTU(d, q) "v_cvt_u32_f32 %" #d ", %" #d "\n"
#define I_RCP(d, q) "v_rcp_f32 %" #d ", %" #d "\n"

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float f[8];
    double r[8];
    for (int i = 0; i < 8; ++i) {
        r[i] = 1.000001 + threadIdx.x * 1e-9 + i;
        f[i] = 1.5f + i + threadIdx.x;
    }
    const float c = 1.0000001f;
    const double c64 = 1.0000001;
    switch (OP) {
#define CASE(n, M) case n: { BODY(M) } break;
        CASE(0, I_AND) CASE(1, I_XOR) CASE(2, I_ADD) CASE(3, I_SHL) CASE(4, I_MIN) CASE(5, I_MIN3) CASE(6, I_LSHLADD)
        CASE(7, I_LSHLOR) CASE(8, I_ANDOR) CASE(9, I_ADD3) CASE(10, I_BFE) CASE(11, I_ALIGN) CASE(12, I_BITOP3)
        CASE(13, I_BCNT) CASE(14, I_MBCNT) CASE(15, I_CNDMASK) CASE(16, I_CMP) CASE(17, I_CMP64) CASE(18, I_ADDC)
        CASE(19, I_MAX3F) CASE(20, I_MAXF) CASE(21, I_ADDF) CASE(22, I_ADDF_ABS) CASE(23, I_MULF) CASE(24, I_FMAF)
        CASE(25, I_FMACF) CASE(26, I_PKADD) CASE(27, I_PKMUL) CASE(28, I_PKFMA) CASE(29, I_PERM) CASE(30, I_MAD24)
        CASE(31, I_MULLO) CASE(32, I_MOV) CASE(33, I_CVTU) CASE(34, I_RCP)
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += f[i] + (float)r[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static const char* kNames[] = {"v_and_b32", "v_xor_b32", "v_add_u32", "v_lshlrev_b32", "v_min_u32", "v_min3_u32", "v_lshl_add_u32",
                               "v_lshl_or_b32", "v_and_or_b32", "v_add3_u32", "v_bfe_u32", "v_alignbit_b32", "v_bitop3_b32",
                               "v_bcnt_u32_b32", "v_mbcnt_lo_u32_b32", "v_cndmask_b32", "v_cmp_ge_u32", "v_cmp_gt_u64", "v_addc_co_u32",
                               "v_max3_f32 |.|", "v_max_f32", "v_add_f32", "v_add_f32 |.|", "v_mul_f32", "v_fma_f32",
                               "v_fmac_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32", "v_perm_b32", "v_mad_u32_u24",
                               "v_mul_lo_u32", "v_mov_b32", "v_cvt_u32_f32", "v_rcp_f32"};

template <int OP>
void run(float* d_out) {
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
        ns[w] = ms * 1e6 / ((double)iters * 64 * waves);
    }
    printf("%-20s %5.2f / %5.2f ns per instruction and SIMD at 4 / 8 waves per SIMD\n", kNames[OP], ns[0], ns[1]);
    if constexpr (OP + 1 < 35) run<OP + 1>(d_out);
}

int main() {
    float* d_out;
    (void)hipMalloc(&d_out, 256 * 8 * 256 * 4);
    run<0>(d_out);
    return 0;
}
