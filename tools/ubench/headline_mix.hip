// Micro-kernel for HISTORY.md (rounds 1-3 text) section 4.1 "route (b)": does a THIRD wave per SIMD pay for the instructions it costs?
//
// frame_rows_pruned_kernel (k_rows_pruned.hip) issues, per wave and unit of 8 windows, 629 packed (v_pk_fma / v_pk_add /
// v_pk_mul) + 192 plain VALU + 150 LDS instructions from ~240 VGPRs: two waves per SIMD.  The 16-lanes-per-window
// layout would halve the data registers (three waves per SIMD, three workgroups of <= 53 KB per CU) at the price of
// +17 % plain VALU (44 v_permlane32_swap + 22 copies per lane and window for stage 6) and twice the LDS transpose
// traffic.  This program issues exactly those two instruction mixes -- independent packed butterfly-like chains over
// the register budget of each layout, plain ops, conflict-free ds_write_b64 / ds_read_b64 in batches of 8 behind one
// s_waitcnt -- with NOTHING else (no barriers, no global memory, no claims), i.e. an upper bound for both, and
// prints the time per unit and SIMD.
//
// build: hipcc --offload-arch=gfx950 -O3 -o headline_mix headline_mix.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

// one "unit": PK packed, PL plain, LD LDS instructions; REGS packed registers (pairs) in rotation
template <int PK, int PL, int LD, int REGS>
__global__ __launch_bounds__(256) void mix(float* out, int units, int lds_floats_per_wave) {
    extern __shared__ float lds[];
    double r[REGS];                                     // a double = one even-aligned VGPR pair
#pragma unroll
    for (int i = 0; i < REGS; ++i) r[i] = (double)(threadIdx.x + i) * 1e-3;
    float p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = threadIdx.x * 0.5f + i;
    const double tw = 0.7071067811865476;
    float* mine = lds + (threadIdx.x >> 6) * lds_floats_per_wave + (threadIdx.x & 63) * 2;   // 8 bytes per lane, no conflicts
    constexpr int ROUNDS = 8;                            // the mix is issued in 8 equal slices per unit
    double l[8];                                        // landing registers of the LDS reads
#pragma unroll
    for (int i = 0; i < 8; ++i) l[i] = 0.0;
    for (int u = 0; u < units; ++u) {
#pragma unroll
        for (int s = 0; s < ROUNDS; ++s) {
            // the slice's LDS traffic first, its arithmetic behind it, one wait at the end: the latency hides
            // behind the VALU work as in the real kernel
#pragma unroll
            for (int i = 0; i < LD / ROUNDS / 2; ++i)
                asm volatile("ds_write_b64 %0, %1" ::"v"((unsigned)(size_t)mine), "v"(r[i % REGS]) : "memory");
#pragma unroll
            for (int i = 0; i < LD / ROUNDS / 2; ++i)
                asm volatile("ds_read_b64 %0, %1" : "=v"(l[i & 7]) : "v"((unsigned)(size_t)mine) : "memory");
#pragma unroll
            for (int i = 0; i < PK / ROUNDS; ++i) {
                const int a = (i * 7 + s) % REGS, b = (i * 7 + s + REGS / 2) % REGS;
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r[a]) : "v"(r[b]), "v"(tw));
                if (i < PL / ROUNDS) asm volatile("v_add_f32 %0, %0, %1" : "+v"(p[i & 7]) : "v"(p[(i + 3) & 7]));
            }
#pragma unroll
            for (int i = PK / ROUNDS; i < PL / ROUNDS; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(p[i & 7]) : "v"(p[(i + 3) & 7]));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    double acc = 0;
#pragma unroll
    for (int i = 0; i < REGS; ++i) acc += r[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += l[i];
    float facc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) facc += p[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)acc + facc;
}

template <int PK, int PL, int LD, int REGS>
double run(const char* name, int wg_per_cu, int lds_bytes_per_wg, float* d_out) {
    const int units = 400, cus = 256;
    auto kern = mix<PK, PL, LD, REGS>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes_per_wg);
    const int blocks = cus * wg_per_cu;
    const int lds_floats_per_wave = lds_bytes_per_wg / 4 / 4;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds_bytes_per_wg, 0, d_out, units, lds_floats_per_wave);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds_bytes_per_wg, 0, d_out, units, lds_floats_per_wave);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern));
    const double waves_per_simd = wg_per_cu;             // 4 waves per workgroup, 4 SIMDs per CU
    const double ns_per_unit_simd = ms * 1e6 / (units * waves_per_simd);
    printf("%-44s VGPRs %3d  waves/SIMD %.0f  LDS/wg %3d KB  %8.1f ns per wave-unit and SIMD  (%.3f ms)\n", name, fa.numRegs,
           waves_per_simd, lds_bytes_per_wg >> 10, ns_per_unit_simd, ms);
    return ns_per_unit_simd;
}

int main() {
    float* d_out;
    hipMalloc(&d_out, 256 * 4 * 256 * 4);
    // today: 629 + 192 + 150 per unit of 8 windows, 2 waves per SIMD (two workgroups of 75.9 KB per CU)
    const double a2 = run<632, 192, 144, 100>("today's mix, 8 lanes per window", 2, 76 << 10, d_out);
    const double a1 = run<632, 192, 144, 100>("  same, one wave per SIMD", 1, 76 << 10, d_out);
    // route (b): +17 % of all lane-instructions as plain VALU (+140 per unit), twice the transpose traffic,
    // half the data registers; three workgroups of 53 KB per CU
    const double b3 = run<632, 336, 288, 56>("route (b) mix, 16 lanes per window", 3, 53 << 10, d_out);
    const double b2 = run<632, 336, 288, 56>("  same, two waves per SIMD", 2, 53 << 10, d_out);
    // the arithmetic alone at three waves (what a free third wave would give)
    const double f3 = run<632, 192, 144, 56>("today's mix if it fitted three waves", 3, 53 << 10, d_out);
    printf("\nper unit and SIMD: today %.0f ns; route (b) %.0f ns (%.2f x today); a third wave for free %.0f ns (%.2f x)\n", a2, b3,
           a2 / b3, f3, a2 / f3);
    // a unit occupies one wave on each of a CU's four SIMDs: 2 M units = 8 M wave-units over 1024 SIMDs
    printf("measured kernel: 16.7 ms per 2 M units (4 waves each) on 1024 SIMDs = %.0f ns per wave-unit and SIMD\n",
           16.7e6 / (4 * 2.0e6 / 1024));
    (void)a1; (void)b2;
    return 0;
}
