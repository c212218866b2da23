// How much dynamic LDS may a kernel ask for, with and without static LDS next to it?  hipcc --offload-arch=gfx950 lds_limit.hip -o lds_limit
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_dyn(float* out) { extern __shared__ float s[]; s[threadIdx.x] = 1.0f; __syncthreads(); out[threadIdx.x] = s[threadIdx.x ^ 1]; }
__global__ void k_mix(float* out) { extern __shared__ float s[]; __shared__ float t[96]; t[threadIdx.x % 96] = 2.0f; s[threadIdx.x] = 1.0f; __syncthreads(); out[threadIdx.x] = s[threadIdx.x ^ 1] + t[3]; }
template <typename K> static void probe(const char* name, K kern) {
    float* d; hipMalloc(&d, 4096);
    for (int bytes = 150 * 1024; bytes <= 164 * 1024; bytes += 256) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        hipError_t l = hipSuccess;
        if (e == hipSuccess) { hipLaunchKernelGGL(kern, dim3(1), dim3(64), bytes, 0, d); l = hipGetLastError(); if (l == hipSuccess) l = hipDeviceSynchronize(); }
        if (e != hipSuccess || l != hipSuccess) { printf("%s: first failure at %d bytes (attr %d, launch %d)\n", name, bytes, (int)e, (int)l); (void)hipGetLastError(); break; }
    }
    hipFree(d);
}
int main() { probe("dynamic only", k_dyn); probe("dynamic + 384 B static", k_mix); hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); printf("sharedMemPerBlock %zu optin %zu\n", p.sharedMemPerBlock, p.sharedMemPerBlockOptin); return 0; }
