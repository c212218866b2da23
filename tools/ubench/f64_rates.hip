// Issue rate of the double-precision instructions the file converter (k_resample.hip) is made of, next to single-precision,
// packed and integer ones: 8 independent chains per wave, 1 / 2 / 4 / 8 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o f64_rates f64_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

// operands: %0..%7 = r[0..7] (64-bit), %8..%15 = f[0..7] (32-bit), %16 = c (64-bit), %17 = cf (32-bit)
#define OPS8(A) A(0, 8) A(1, 9) A(2, 10) A(3, 11) A(4, 12) A(5, 13) A(6, 14) A(7, 15)
#define REP8(x) x x x x x x x x
#define BODY(A)                                                                                                         \
    for (int it = 0; it < iters; ++it) {                                                                                \
        REP8(asm volatile(OPS8(A)                                                                                       \
                          : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), \
                            "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7])  \
                          : "v"(c), "v"(cf));)                                                                          \
    }

#define A_ADD(d, s) "v_add_f64 %" #d ", %" #d ", %16\n"
#define A_FMA(d, s) "v_fma_f64 %" #d ", %" #d ", %16, %" #d "\n"
#define A_MUL(d, s) "v_mul_f64 %" #d ", %" #d ", %16\n"
#define A_TRUNC(d, s) "v_trunc_f64 %" #d ", %" #d "\n"
#define A_FRACT(d, s) "v_fract_f64 %" #d ", %" #d "\n"
#define A_FLOOR(d, s) "v_floor_f64 %" #d ", %" #d "\n"
#define A_CVTU(d, s) "v_cvt_u32_f64 %" #s ", %" #d "\n"
#define A_CVTD(d, s) "v_cvt_f64_u32 %" #d ", %" #s "\n"
#define A_CVTF(d, s) "v_cvt_f64_f32 %" #d ", %" #s "\n"
#define A_F32(d, s) "v_fma_f32 %" #s ", %" #s ", %17, %" #s "\n"
#define A_MOV(d, s) "v_mov_b32 %" #s ", %17\n"
#define A_PK(d, s) "v_pk_fma_f32 %" #d ", %" #d ", %16, %" #d "\n"
#define A_INT(d, s) "v_lshl_add_u32 %" #s ", %" #s ", 1, %17\n"

template <int OP>
__global__ __launch_bounds__(256) void k(double* out, int iters, long long* cyc) {
    double r[8];
    float f[8];
    for (int i = 0; i < 8; ++i) {
        r[i] = 1.000001 + threadIdx.x * 1e-9 + i;
        f[i] = 1.5f + i + threadIdx.x;
    }
    const double c = 1.0000001;
    const float cf = 1.0000001f;
    const long long t0 = __builtin_readcyclecounter();
    if (OP == 0) { BODY(A_ADD) }
    if (OP == 1) { BODY(A_FMA) }
    if (OP == 2) { BODY(A_MUL) }
    if (OP == 3) { BODY(A_TRUNC) }
    if (OP == 4) { BODY(A_FRACT) }
    if (OP == 5) { BODY(A_FLOOR) }
    if (OP == 6) { BODY(A_CVTU) }
    if (OP == 7) { BODY(A_CVTD) }
    if (OP == 8) { BODY(A_CVTF) }
    if (OP == 9) { BODY(A_F32) }
    if (OP == 10) { BODY(A_MOV) }
    if (OP == 11) { BODY(A_PK) }
    if (OP == 12) { BODY(A_INT) }
    const long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += r[i] + f[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int OP>
void run(const char* name, double* d_out, long long* d_cyc) {
    const int iters = 2000;
    for (int waves = 1; waves <= 8; waves *= 2) {            // blocks of 4 waves: 1, 2, 4, 8 waves per SIMD
        hipLaunchKernelGGL(k<OP>, dim3(256 * waves), dim3(256), 0, 0, d_out, iters, d_cyc);
        (void)hipDeviceSynchronize();
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(256 * waves), dim3(256), 0, 0, d_out, iters, d_cyc);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        long long cyc;
        (void)hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
        const double n = (double)iters * 64;                 // instructions per wave
        printf("%-14s %d wave(s)/SIMD: %6.2f clock-counter ticks per instruction (first wave), %7.3f ms = %5.2f ns per instruction and SIMD\n",
               name, waves, (double)cyc / n, ms, ms * 1e6 / (n * waves));
    }
}

int main() {
    double* d_out;
    long long* d_cyc;
    (void)hipMalloc(&d_out, 256 * 8 * 256 * 8);
    (void)hipMalloc(&d_cyc, 8);
    run<9>("v_fma_f32", d_out, d_cyc);
    run<10>("v_mov_b32", d_out, d_cyc);
    run<12>("v_lshl_add_u32", d_out, d_cyc);
    run<11>("v_pk_fma_f32", d_out, d_cyc);
    run<0>("v_add_f64", d_out, d_cyc);
    run<1>("v_fma_f64", d_out, d_cyc);
    run<2>("v_mul_f64", d_out, d_cyc);
    run<3>("v_trunc_f64", d_out, d_cyc);
    run<4>("v_fract_f64", d_out, d_cyc);
    run<5>("v_floor_f64", d_out, d_cyc);
    run<6>("v_cvt_u32_f64", d_out, d_cyc);
    run<7>("v_cvt_f64_u32", d_out, d_cyc);
    run<8>("v_cvt_f64_f32", d_out, d_cyc);
    return 0;
}
