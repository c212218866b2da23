// What an exec-masked global_load_dwordx4 costs on gfx950 as a function of the lanes that are switched on -- the feeder
// loads of k_sliding.hip's step loop (two per step, about ten of 64 lanes active).  Every wave walks its own 256 KB window
// (L2-resident after the first lap), 32 bytes per lane and iteration, with `valu` independent v_bitop3 between the loads.
//   build: hipcc --offload-arch=gfx950 -O3 -o vmem_rates vmem_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int VALU, int LOADS>
__global__ __launch_bounds__(1024) void k(const uint4* __restrict__ buf, unsigned* out, int iters, unsigned long long mask, unsigned stride16) {
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    // lane l reads a stream of its own: 4 KB apart, 32 B per iteration (like a feeder lane)
    const uint4* p = buf + (size_t)wave * 16384 + (size_t)lane * stride16;
    u32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
    unsigned f[8];
    for (int i = 0; i < 8; ++i) f[i] = lane + i;
    const unsigned c = 0x12345u + lane;
    for (int it = 0; it < iters; ++it) {
        const uint4* q = p + 2 * (it & 3);
        if (LOADS) {
            unsigned long long saved;
            asm volatile("s_mov_b64 %2, exec\n\t"
                         "s_mov_b64 exec, %4\n\t"
                         "global_load_dwordx4 %0, %3, off\n\t"
                         "global_load_dwordx4 %1, %3, off offset:16\n\t"
                         "s_mov_b64 exec, %2"
                         : "+v"(a), "+v"(b), "=&s"(saved)
                         : "v"(q), "s"(mask)
                         : "memory");
        }
#pragma unroll
        for (int r = 0; r < VALU / 8; ++r)
            asm volatile("v_bitop3_b32 %0, %0, %8, %0 bitop3:0x96\n v_bitop3_b32 %1, %1, %8, %1 bitop3:0x96\n"
                         "v_bitop3_b32 %2, %2, %8, %2 bitop3:0x96\n v_bitop3_b32 %3, %3, %8, %3 bitop3:0x96\n"
                         "v_bitop3_b32 %4, %4, %8, %4 bitop3:0x96\n v_bitop3_b32 %5, %5, %8, %5 bitop3:0x96\n"
                         "v_bitop3_b32 %6, %6, %8, %6 bitop3:0x96\n v_bitop3_b32 %7, %7, %8, %7 bitop3:0x96\n"
                         : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7])
                         : "v"(c));
        if (LOADS) asm volatile("s_waitcnt vmcnt(2)" : "+v"(a), "+v"(b));
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b));
    unsigned s = a.x + b.y;
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int VALU, int LOADS>
double run(const uint4* d_buf, unsigned* d_out, int threads, unsigned long long mask, unsigned stride16) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<VALU, LOADS>), dim3(256), dim3(threads), 0, 0, d_buf, d_out, iters, mask, stride16);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<VALU, LOADS>), dim3(256), dim3(threads), 0, 0, d_buf, d_out, iters, mask, stride16);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / iters;      // ns per iteration of every wave of a CU (they run side by side)
}

static unsigned long long lanes(int n, int spread) {
    unsigned long long m = 0;
    for (int i = 0; i < n; ++i) m |= 1ull << ((i * spread) & 63);
    return m;
}

int main() {
    uint4* d_buf;
    unsigned* d_out;
    const size_t bytes = (size_t)256 * 16 * 16384 * 16;      // 256 KB per wave
    (void)hipMalloc(&d_buf, bytes);
    (void)hipMemset(d_buf, 1, bytes);
    (void)hipMalloc(&d_out, 256 * 1024 * 4);
    printf("ns per iteration (2 masked global_load_dwordx4 + VALU v_bitop3) of all waves of a CU, one workgroup per CU\n");
    for (int threads = 512; threads <= 1024; threads *= 2) {
        printf("-- %d waves per CU\n", threads / 64);
        printf("   VALU only:            v0 %7.1f  v64 %7.1f\n", run<0, 0>(d_buf, d_out, threads, 0, 256), run<64, 0>(d_buf, d_out, threads, 0, 256));
        const int ns[] = {0, 1, 4, 10, 16, 32, 64};
        for (int n : ns) {
            const unsigned long long m = n == 64 ? ~0ull : lanes(n, 6);
            printf("   %2d lanes, 4 KB apart:  v0 %7.1f  v64 %7.1f    2 B apart (one line): v0 %7.1f\n", n, run<0, 1>(d_buf, d_out, threads, m, 256),
                   run<64, 1>(d_buf, d_out, threads, m, 256), run<0, 1>(d_buf, d_out, threads, m, 0));
        }
    }
    return 0;
}
