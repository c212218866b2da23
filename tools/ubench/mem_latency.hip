// Load-to-use latency of HBM-resident data as a function of how many waves are asking: every wave runs a chain of
// dependent 16-byte loads through its own 8 MB region (a new 4 KB-aligned line every time, no reuse), `lanes` lanes active.
//   build: hipcc --offload-arch=gfx950 -O3 -o mem_latency mem_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(64) void chase(const uint4* __restrict__ buf, size_t region16, int steps, unsigned long long* out, int lanes) {
    const unsigned lane = threadIdx.x & 63u;
    const uint4* base = buf + (size_t)blockIdx.x * region16;
    unsigned idx = lane * 8u;                          // lanes 128 B apart
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned acc = 0;
    if ((int)lane < lanes) {
        for (int i = 0; i < steps; ++i) {
            const uint4 v = base[idx];
            acc += v.x;
            idx = (idx + 4096u / 16u * 37u + (v.y & 1u)) % (unsigned)region16;     // depends on the data: a true chain
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) out[blockIdx.x] = t1 - t0;
    if (acc == 0x12345678u) out[blockIdx.x] = 0;
}

int main() {
    const size_t region = 8u << 20, max_wg = 4096;
    uint4* d;
    unsigned long long* d_out;
    if (hipMalloc(&d, region * max_wg) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(d, 0, region * max_wg);
    (void)hipMalloc(&d_out, max_wg * 8);
    const int steps = 200;
    printf("dependent 16-byte loads from HBM (no reuse), %d per wave; latency = wave time / loads\n", steps);
    for (int lanes : {1, 16, 64}) {
        for (int grid : {1, 8, 32, 256, 1024, 4096}) {
            hipLaunchKernelGGL(chase, dim3(grid), dim3(64), 0, 0, d, region / 16, steps, d_out, lanes);
            (void)hipDeviceSynchronize();
            hipLaunchKernelGGL(chase, dim3(grid), dim3(64), 0, 0, d, region / 16, steps, d_out, lanes);
            (void)hipDeviceSynchronize();
            std::vector<unsigned long long> h(grid);
            (void)hipMemcpy(h.data(), d_out, grid * 8, hipMemcpyDeviceToHost);
            double sum = 0, mx = 0;
            for (auto v : h) { sum += (double)v; mx = v > mx ? (double)v : mx; }
            printf("  %2d lanes, %4d waves on the chip: mean %6.0f ns, slowest wave %6.0f ns per load\n", lanes, grid, sum / grid / steps * 10.0,
                   mx / steps * 10.0);
        }
    }
    return 0;
}
