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

// One output sample of the band-limited models.  Same operations on the same values as audiofile.cpp's loop, in
// the same order; only the bookkeeping differs: the tap index runs as a double (k + 1.0 is exact), the table index
// is a 32-bit conversion (|k - pos| * coord <= 24 * 2048 + 1), and taps whose input lies outside the file are split
// off into their own loops instead of being tested one by one.  A tap the table does not cover contributes
// +0.0 to both sums (the host skips it; neither sum can be -0.0, so the results are the same bits).
__device__ __forceinline__ float sinc_sample(const float* __restrict__ in, uint64_t n_in, double ratio, double scale,
                                             double half, int res, const double* __restrict__ table, uint64_t table_n,
                                             uint64_t n) {
    const double pos = (double)n * ratio;
    const long k0 = (long)ceil(pos - half), k1 = (long)floor(pos + half);
    const double coord = (double)res / scale;                     // table points per input sample, as on the host
    const uint32_t tn = table_n > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)table_n;
    double acc = 0.0, wsum = 0.0;
    bool ok = true;
    auto weight = [&](double kd) -> double {
        const double t = fabs(kd - pos) * coord;                  // table coordinate
        const uint32_t i = (uint32_t)t;
        ok = i + 1u < tn && t < 4294967040.0;
        const uint32_t ic = ok ? i : 0u;
        const double a = table[ic], b = table[ic + 1u];
        const double w = a + (b - a) * (t - (double)i);
        return ok ? w : 0.0;
    };
    // taps left of the file, inside it, right of it -- ascending k throughout
    const long in_lo = k0 > 0 ? k0 : 0;
    const long in_hi = k1 < (long)n_in - 1 ? k1 : (long)n_in - 1;
    long k = k0;
    double kd = (double)k0;
    for (; k <= k1 && k < in_lo; ++k, kd += 1.0) wsum += weight(kd);
    for (; k <= in_hi; ++k, kd += 1.0) {
        const double w = weight(kd);
        wsum += w;
        const float x = in[(uint64_t)k];
        acc += w * (ok ? (double)x : 0.0);                        // an uncovered tap must not meet an inf / NaN sample
    }
    for (; k <= k1; ++k, kd += 1.0) wsum += weight(kd);
    return (float)(wsum != 0.0 ? acc / wsum : 0.0);
}

__device__ __forceinline__ float linear_sample(const float* __restrict__ in, uint64_t n_in, double ratio, uint64_t n) {
    const double pos = (double)n * ratio;
    const uint64_t k = (uint64_t)pos;
    const double f = pos - (double)k;
    const double a = k < n_in ? (double)in[k] : 0.0, b = k + 1 < n_in ? (double)in[k + 1] : 0.0;
    return (float)(a * (1.0 - f) + b * f);
}

__global__ __launch_bounds__(kThreads) void resample_sinc_kernel(const float* __restrict__ in, uint64_t n_in, double ratio,
                                                                 double scale, double half, int res,
                                                                 const double* __restrict__ table, uint64_t table_n,
                                                                 float* __restrict__ out, uint64_t n_out) {
    const uint64_t n = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    if (n >= n_out) return;
    out[n] = sinc_sample(in, n_in, ratio, scale, half, res, table, table_n, n);
}

__global__ __launch_bounds__(kThreads) void resample_linear_kernel(const float* __restrict__ in, uint64_t n_in, double ratio,
                                                                   float* __restrict__ out, uint64_t n_out) {
    const uint64_t n = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    if (n >= n_out) return;
    out[n] = linear_sample(in, n_in, ratio, n);
}

// every file of a batch in one launch: blockIdx.y = file, blockIdx.x walks its output samples; a file whose rate is
// the processing rate is copied
__global__ __launch_bounds__(kThreads) void resample_batch_kernel(const FileDesc* __restrict__ files, const float* __restrict__ decoded,
                                                                  int res, const double* __restrict__ table, uint64_t table_n,
                                                                  float* __restrict__ pcm) {
    const FileDesc f = files[blockIdx.y];
    const float* in = decoded + f.dec_off + f.first;
    float* out = pcm + f.out_off;
    for (uint64_t n = (uint64_t)blockIdx.x * kThreads + threadIdx.x; n < f.n_write; n += (uint64_t)gridDim.x * kThreads) {
        float v;
        if (f.copy) v = in[n];
        else if (f.mode == 2) v = linear_sample(in, f.n_in, f.ratio, n);
        else v = sinc_sample(in, f.n_in, f.ratio, f.scale, f.half, res, table, table_n, n);
        out[n] = v;
    }
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

// d_files: n_files descriptors (all with the same converter mode, hence one table); max_out: the largest n_write
hipError_t launch_resample_batch(const FileDesc* d_files, uint32_t n_files, uint64_t max_out, const float* d_decoded, int res,
                                 const double* d_table, uint64_t table_n, float* d_pcm, hipStream_t stream) {
    if (n_files == 0 || max_out == 0) return hipSuccess;
    if (n_files > 65535u) return hipErrorInvalidValue;
    uint64_t bx = (max_out + kThreads - 1) / kThreads;
    if (bx > 65535) bx = 65535;
    hipLaunchKernelGGL(resample_batch_kernel, dim3((uint32_t)bx, n_files), dim3(kThreads), 0, stream, d_files, d_decoded, res,
                       d_table, table_n, d_pcm);
    return hipGetLastError();
}

}  // namespace lbad
