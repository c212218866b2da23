// Micro-benchmark: issue cost of plain / packed f32 VALU ops on gfx950, by waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void k(float* out, long long* cyc, int iters) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
    const float w = 1.0001f, c = 0.5f;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {   // 16 independent v_fma_f32 per round x 8 rounds
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(w), "v"(c));
        } else if (MODE == 1) {   // 8 independent v_pk_fma_f32 (2 floats each) x 8 rounds
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 16; i += 2)
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*reinterpret_cast<double*>(&a[i])) : "v"(*reinterpret_cast<const double*>(&a[14])), "v"(*reinterpret_cast<const double*>(&a[12])));
        } else if (MODE == 2) {   // dependent chain of v_fma_f32
#pragma unroll
            for (int r = 0; r < 128; ++r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(w), "v"(c));
        } else if (MODE == 3) {   // 16 independent v_add_f32
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        } else if (MODE == 4) {   // 2 interleaved dependent chains
#pragma unroll
            for (int r = 0; r < 64; ++r) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(w), "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[1]) : "v"(w), "v"(c));
            }
        } else if (MODE == 5) {   // v_fmac with literal constant (VOP2)
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32 %0, 0x3f7fff58, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads, float* d_out, long long* d_cyc) {
    const int iters = 2000, blocks = 256;   // one block per CU
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d_out, d_cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d_out, d_cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> c(blocks);
    hipMemcpy(c.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : c) avg += v; avg /= blocks;
    const double n_inst = 128.0 * iters * (MODE == 1 ? 0.5 : 1.0);   // wave-instructions per wave
    const int waves_per_simd = threads / 256;
    printf("%-28s waves/SIMD=%d  counter-ticks/inst/wave=%7.3f  wall ns/inst/SIMD=%7.3f  (%.3f ms)\n", name, waves_per_simd,
           avg / n_inst, ms * 1e6 / (n_inst * waves_per_simd), ms);
}

int main() {
    float* d_out; long long* d_cyc;
    hipMalloc(&d_out, 256 * 1024 * 4); hipMalloc(&d_cyc, 256 * 8);
    for (int threads : {256, 512, 768, 1024}) {
        run<0>("v_fma_f32 x16 indep", threads, d_out, d_cyc);
        run<1>("v_pk_fma_f32 x8 indep", threads, d_out, d_cyc);
        run<2>("v_fma_f32 dependent", threads, d_out, d_cyc);
        run<4>("v_fma_f32 2 chains", threads, d_out, d_cyc);
        run<3>("v_add_f32 x16 indep", threads, d_out, d_cyc);
        run<5>("v_fmac_f32 literal x16", threads, d_out, d_cyc);
    }
    return 0;
}
