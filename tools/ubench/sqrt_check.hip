// Does __fsqrt_rn / sqrtf / __fdiv_rn on the device match the host's correctly rounded results?
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(float* a, float* b, float* c, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        a[i] = __fsqrt_rn((float)(i + 1));
        b[i] = sqrtf((float)(i + 1));
        c[i] = __fdiv_rn(100.0f, __fsqrt_rn((float)(i + 1)));
    }
}
int main() {
    const int n = 1 << 20;
    float *a, *b, *c;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&c, n * 4);
    k<<<(n + 255) / 256, 256>>>(a, b, c, n);
    std::vector<float> ha(n), hb(n), hc(n);
    hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hc.data(), c, n * 4, hipMemcpyDeviceToHost);
    int bad_a = 0, bad_b = 0, bad_c = 0;
    for (int i = 0; i < n; ++i) {
        const float w = sqrtf((float)(i + 1));
        if (ha[i] != w) { if (bad_a < 5) printf("__fsqrt_rn(%d) = %a, host %a\n", i + 1, ha[i], w); ++bad_a; }
        if (hb[i] != w) ++bad_b;
        if (hc[i] != 100.0f / w) ++bad_c;
    }
    printf("__fsqrt_rn mismatches %d, sqrtf mismatches %d, 100/sqrt mismatches %d of %d\n", bad_a, bad_b, bad_c, n);
    return 0;
}
