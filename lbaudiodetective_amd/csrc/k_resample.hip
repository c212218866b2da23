// k_resample.hip -- the sample-rate converter of the file front end on the device.
//
// Upstream hands the file to ExtAudioFile with a client format at the processing rate (LBAudioDetective.m:224-237)
// and Apple's converter does the rest; audiofile.cpp documents the three stand-ins.  The arithmetic here is
// resample()'s, operation for operation in double precision (no contraction, IEEE division), so a file converted on
// the device has the samples the host function returns, bit for bit -- only 100 x sooner: one thread per output
// sample, the 385 taps of the long kernel per thread, the input (1.6 MB for a 9 s file) read through L2.
#include "internal.hpp"

namespace lbad {
namespace {

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void resample_sinc_kernel(const float* __restrict__ in, uint64_t n_in, double ratio,
                                                                 double scale, double half, int res,
                                                                 const double* __restrict__ table, uint64_t table_n,
                                                                 float* __restrict__ out, uint64_t n_out) {
    const uint64_t n = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    if (n >= n_out) return;
    const double pos = (double)n * ratio;
    const long k0 = (long)ceil(pos - half), k1 = (long)floor(pos + half);
    double acc = 0.0, wsum = 0.0;
    for (long k = k0; k <= k1; ++k) {
        const double t = fabs(((double)k - pos) / scale) * res;   // table coordinate
        const uint64_t i = (uint64_t)t;
        if (i + 1 >= table_n) continue;
        const double w = table[i] + (table[i + 1] - table[i]) * (t - (double)i);
        wsum += w;
        if (k >= 0 && (uint64_t)k < n_in) acc += w * (double)in[(uint64_t)k];
    }
    out[n] = (float)(wsum != 0.0 ? acc / wsum : 0.0);
}

__global__ __launch_bounds__(kThreads) void resample_linear_kernel(const float* __restrict__ in, uint64_t n_in, double ratio,
                                                                   float* __restrict__ out, uint64_t n_out) {
    const uint64_t n = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    if (n >= n_out) return;
    const double pos = (double)n * ratio;
    const uint64_t k = (uint64_t)pos;
    const double f = pos - (double)k;
    const double a = k < n_in ? (double)in[k] : 0.0, b = k + 1 < n_in ? (double)in[k + 1] : 0.0;
    out[n] = (float)(a * (1.0 - f) + b * f);
}

}  // namespace

hipError_t launch_resample(const float* d_in, uint64_t n_in, uint32_t mode, double ratio, double scale, double half,
                           int res, const double* d_table, uint64_t table_n, float* d_out, uint64_t n_out,
                           hipStream_t stream) {
    if (n_out == 0) return hipSuccess;
    const uint64_t blocks = (n_out + kThreads - 1) / kThreads;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    if (mode == 2)
        hipLaunchKernelGGL(resample_linear_kernel, dim3((uint32_t)blocks), dim3(kThreads), 0, stream, d_in, n_in, ratio, d_out, n_out);
    else
        hipLaunchKernelGGL(resample_sinc_kernel, dim3((uint32_t)blocks), dim3(kThreads), 0, stream, d_in, n_in, ratio, scale,
                           half, res, d_table, table_n, d_out, n_out);
    return hipGetLastError();
}

}  // namespace lbad
