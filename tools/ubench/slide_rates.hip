// Issue cost of the instruction forms k_sliding.hip's step loop is made of that tools/ubench/valu_rates.hip does not cover:
// v_bitop3_b32 with a SCALAR operand, the accumulating v_bcnt, v_mov_b32 with wave_shl:1 / wave_shr:1 / row_shr:1 DPP.
// 8 independent chains per wave, 4 and 8 waves per SIMD.   build: hipcc --offload-arch=gfx950 -O3 -o slide_rates slide_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define OPS8(A) A(0) A(1) A(2) A(3) A(4) A(5) A(6) A(7)
#define REP8(x) x x x x x x x x
#define BODY(A)                                                                                                          \
    for (int it = 0; it < iters; ++it) {                                                                                 \
        REP8(asm volatile(OPS8(A)                                                                                        \
                          : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) \
                          : "v"(c), "s"(sc));)                                                                           \
    }
#define I_BITOP3_V(d) "v_bitop3_b32 %" #d ", %" #d ", %8, %" #d " bitop3:0x96\n"
#define I_BITOP3_S(d) "v_bitop3_b32 %" #d ", %" #d ", %8, %9 bitop3:0x96\n"
#define I_AND_S(d) "v_and_b32 %" #d ", %9, %" #d "\n"
#define I_BCNT_ACC(d) "v_bcnt_u32_b32 %" #d ", %8, %" #d "\n"
#define I_DPP_WSHL(d) "v_mov_b32_dpp %" #d ", %" #d " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define I_DPP_WSHR(d) "v_mov_b32_dpp %" #d ", %" #d " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define I_DPP_RSHR(d) "v_mov_b32_dpp %" #d ", %" #d " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define I_DPP_WSHL_FROM(d) "v_mov_b32_dpp %" #d ", %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define I_PERMLANE(d) "v_mov_b32 %" #d ", %" #d "\n"

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, unsigned sc) {
    unsigned f[8];
    for (int i = 0; i < 8; ++i) f[i] = 15u + i + threadIdx.x;
    const unsigned c = 0x12345u + threadIdx.x;
    switch (OP) {
#define CASE(n, M) case n: { BODY(M) } break;
        CASE(0, I_BITOP3_V) CASE(1, I_BITOP3_S) CASE(2, I_AND_S) CASE(3, I_BCNT_ACC) CASE(4, I_DPP_WSHL) CASE(5, I_DPP_WSHR)
        CASE(6, I_DPP_RSHR) CASE(7, I_DPP_WSHL_FROM) CASE(8, I_PERMLANE)
    }
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
static const char* kNames[] = {"v_bitop3_b32 vvv", "v_bitop3_b32 vvS", "v_and_b32 S", "v_bcnt accumulate", "v_mov_dpp wave_shl:1 (chain)",
                               "v_mov_dpp wave_shr:1 (chain)", "v_mov_dpp row_shr:1 (chain)", "v_mov_dpp wave_shl:1 (indep)", "v_mov_b32 v,v"};
template <int OP>
void run(unsigned* d_out) {
    const int iters = 1000;
    double ns[2];
    int w = 0;
    for (int waves = 4; waves <= 8; waves *= 2, ++w) {
        hipLaunchKernelGGL(k<OP>, dim3(256 * waves), dim3(256), 0, 0, d_out, iters, 0x55u);
        (void)hipDeviceSynchronize();
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(256 * waves), dim3(256), 0, 0, d_out, iters, 0x55u);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        ns[w] = ms * 1e6 / ((double)iters * 64 * waves);
    }
    printf("%-30s %5.2f / %5.2f ns per instruction and SIMD at 4 / 8 waves per SIMD\n", kNames[OP], ns[0], ns[1]);
    if constexpr (OP + 1 < 9) run<OP + 1>(d_out);
}
int main() {
    unsigned* d_out;
    (void)hipMalloc(&d_out, 256 * 8 * 256 * 4);
    run<0>(d_out);
    return 0;
}
