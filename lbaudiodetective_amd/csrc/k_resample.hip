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

// Band-limited models with the kernel table staged through LDS.  The table coordinate of tap j (counted from the
// output's own first tap k0) is |j - half + f| * coord with f = k0 - (pos - half) in [0, 1): it depends on the output
// only through its phase f.  When the file rate is (nearly) a multiple of the processing rate -- 44.1 kHz -> 5512 Hz:
// ratio 8.0007 -- the phases of 256 consecutive outputs lie within 0.19 of each other, so for every tap the whole
// block reads a run of ~50 consecutive table points.  The block stages those runs for kTapGroup taps at a time
// (rows of kRowLen doubles) and every lane takes its two points from LDS instead of 16 bytes through L1 per tap
// (23 GB per sixty files, what bounded the kernel).  Same values, same operations, same order as sinc_sample;
// a lane whose index falls outside the staged run (or a block whose phases are spread wider) reads the table itself.
constexpr int kRowLen = 64;
constexpr int kTapGroup = 48;
// The block's input samples go through LDS as well: at a ratio of 8 the lanes of a wave read samples 32 bytes apart
// (16 cache lines per load instruction and tap).  One pad word per 64 keeps lanes 8 samples apart on distinct banks.
constexpr int kInMax = 3072;

__device__ __forceinline__ float sinc_sample_tiled(const float* __restrict__ in, uint64_t n_in, double ratio, double scale,
                                                   double half, int res, const double* __restrict__ table, uint64_t table_n,
                                                   uint64_t n, bool active, double (*s_rows)[kRowLen], uint32_t* s_lo,
                                                   uint32_t* s_stat, float* s_in) {
    const double pos = (double)n * ratio;
    const long k0 = (long)ceil(pos - half), k1 = (long)floor(pos + half);
    const double coord = (double)res / scale;
    const uint32_t tn = table_n > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)table_n;
    // phase spread and tap count of the block (fixed-point phases through LDS atomics: the bounds only place the
    // staged runs, the margins below absorb the rounding)
    const double fr = (double)k0 - (pos - half);
    __syncthreads();                                          // s_stat is reused by consecutive blocks
    if (threadIdx.x == 0) {
        s_stat[0] = 0xFFFFFFFFu;
        s_stat[1] = 0u;
        s_stat[2] = 0u;
        s_stat[3] = 0x7FFFFFFFu;                              // smallest first tap (as int)
        s_stat[4] = 0x80000001u;                              // largest last tap (as int)
        s_stat[5] = 0xFFFFFFFFu;                              // smallest tap count
    }
    __syncthreads();
    if (active) {
        const double fc = fr < 0.0 ? 0.0 : (fr > 1.0 ? 1.0 : fr);
        const uint32_t q = (uint32_t)(fc * 1073741824.0);     // 2^30
        atomicMin(&s_stat[0], q);
        atomicMax(&s_stat[1], q + 1u);
        atomicMax(&s_stat[2], (uint32_t)(k1 - k0 + 1));
        atomicMin(&s_stat[5], (uint32_t)(k1 - k0 + 1));
        atomicMin(reinterpret_cast<int*>(&s_stat[3]), (int)(k0 < -2147483647L ? -2147483647L : (k0 > 2147483647L ? 2147483647L : k0)));
        atomicMax(reinterpret_cast<int*>(&s_stat[4]), (int)(k1 < -2147483647L ? -2147483647L : (k1 > 2147483647L ? 2147483647L : k1)));
    }
    __syncthreads();
    const double fmin = (double)s_stat[0] * (1.0 / 1073741824.0), fmax = (double)s_stat[1] * (1.0 / 1073741824.0);
    const int n_taps = (int)s_stat[2];
    if (n_taps == 0) return 0.0f;                             // no active lane in the block
#if defined(LBAD_EXP_FORCE_PLAIN)
    const bool tiled = false;
#elif defined(LBAD_EXP_FORCE_TILED)
    const bool tiled = true;
#else
    const bool tiled = (fmax - fmin) * coord + 6.0 <= (double)kRowLen;
#endif
    if (!tiled) {                                             // phases too far apart: plain reads
        return active ? sinc_sample(in, n_in, ratio, scale, half, res, table, table_n, n) : 0.0f;
    }
    // the input samples every tap of the block reads
    const long kbase = (long)(int)s_stat[3], kend = (long)(int)s_stat[4];
    const bool stage_in = n_in < 0x7FFFFFFFull && kend - kbase + 1 <= (long)kInMax;
    if (stage_in) {
        for (long p = threadIdx.x; p <= kend - kbase; p += kThreads) {
            const long kk = kbase + p;
            s_in[p + (p >> 6)] = kk >= 0 && (uint64_t)kk < n_in ? in[(uint64_t)kk] : 0.0f;
        }
    }
    double acc = 0.0, wsum = 0.0;
    double kd = (double)k0;
    long k = k0;
    uint32_t worst = 0;
    const int min_taps = (int)s_stat[5];
    // coord >= 1 table point per input sample and half * coord < 2^31: what the untested rows below rely on
    const bool fast_ok = stage_in && coord >= 1.0 && half * coord < 2147483648.0 && kend - kbase < 0x7FFFFFFFl;
    // one tap with every test (the host loop's tap, audiofile.cpp)
    auto general_tap = [&](int r) {
        if (!active || k > k1) return;
        const double t = fabs(kd - pos) * coord;              // table coordinate
        const uint32_t i = (uint32_t)t;
        const bool ok = i + 1u < tn && t < 4294967040.0;
        const uint32_t idx = i - s_lo[r];
        double ta, tb;
        if (idx < (uint32_t)(kRowLen - 1)) {                   // unsigned: also false when i < s_lo[r]
            ta = s_rows[r][idx];
            tb = s_rows[r][idx + 1u];
        } else {
            const uint32_t ic = ok ? i : 0u;
            ta = table[ic];
            tb = table[ic + 1u];
        }
        const double w = ok ? ta + (tb - ta) * (t - (double)i) : 0.0;
        wsum += w;
        const bool inside = k >= 0 && (uint64_t)k < n_in;
        const long q = k - kbase;
        const float x = stage_in ? s_in[q + (q >> 6)] : (inside ? in[(uint64_t)k] : 0.0f);
        acc += w * (ok && inside ? (double)x : 0.0);           // + (+-0.0) leaves acc as it is (acc is never -0.0)
    };
    for (int g0 = 0; g0 < n_taps; g0 += kTapGroup) {
        __syncthreads();
        if (threadIdx.x < kTapGroup) {                        // first staged point of row r: 2 points below the smallest index
            const double a = (double)(g0 + (int)threadIdx.x) - half + fmin, b = (double)(g0 + (int)threadIdx.x) - half + fmax;
            const double tmin = a >= 0.0 ? a * coord : (b <= 0.0 ? -b * coord : 0.0);
            const double lo = floor(tmin) - 2.0;
            s_lo[threadIdx.x] = lo > 0.0 ? (lo < 4294967000.0 ? (uint32_t)lo : 0xFFFFFF00u) : 0u;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < kTapGroup * kRowLen; e += kThreads) {
            const int r = e / kRowLen, c = e % kRowLen;
            const uint64_t i = (uint64_t)s_lo[r] + (uint64_t)c;
            s_rows[r][c] = i < table_n ? table[i] : 0.0;
        }
        __syncthreads();
        const int rows = n_taps - g0 < kTapGroup ? n_taps - g0 : kTapGroup;
        // Rows every lane of the block treats alike -- not a lane's first tap, not one of its last two, the inputs staged,
        // the table coordinate far below 2^32 -- run without a test per tap: such a tap lies inside its lane's span
        // (k0 < k < k1), so the table covers it (|k - pos| <= half - 1: a whole input sample = coord >= 1 table points
        // short of the end), a sample outside the file was staged as +0.0, and t - (double)i is v_fract_f64 (the exact
        // difference either way).  The one thing checked is that the two table points came from the staged run: the
        // largest offset met is kept, and a lane that left the run recomputes its sample the plain way afterwards.
        int r = 0;
        if (fast_ok) {
            const uint32_t my_lo = s_lo[(threadIdx.x & 63u) < (uint32_t)kTapGroup ? (threadIdx.x & 63u) : 0u];
            const int r_begin = g0 == 0 ? 1 : 0;
            const int r_end = min_taps - 2 - g0 < rows ? min_taps - 2 - g0 : rows;       // taps g0 + r <= min_taps - 3
            for (; r < r_begin && r < rows; ++r, ++k, kd += 1.0) general_tap(r);
            uint32_t q = (uint32_t)(k - kbase);
            for (; r < r_end; ++r, kd += 1.0, ++q) {             // (unrolling by hand changes nothing: the loop is issue-bound)
                const double t = fabs(kd - pos) * coord;
                const uint32_t i = (uint32_t)t;
                const uint32_t idx = i - (uint32_t)__builtin_amdgcn_readlane((int)my_lo, r);
                worst = idx > worst ? idx : worst;
                const double* row = &s_rows[r][idx < (uint32_t)(kRowLen - 1) ? idx : 0u];
                const double ta = row[0], tb = row[1];
                const double w = ta + (tb - ta) * __builtin_amdgcn_fract(t);
                wsum += w;
                acc += w * (double)s_in[q + (q >> 6)];
            }
            k = kbase + (long)q;
        }
        for (; r < rows; ++r, ++k, kd += 1.0) general_tap(r);
    }
    if (worst >= (uint32_t)(kRowLen - 1))                     // (never seen: the runs are placed with two points to spare)
        return active ? sinc_sample(in, n_in, ratio, scale, half, res, table, table_n, n) : 0.0f;
    return (float)(wsum != 0.0 ? acc / wsum : 0.0);
}

// Rational position (audiofile.hpp): output n = the phase's ready-made weights against the inputs around (n p) div q.
// One thread per output sample; consecutive outputs have consecutive phases (p mod q apart: 1 at 44.1 kHz -> 5512 Hz), so
// the block reads its weights as contiguous runs of the tap-major table (L2-resident: 4.2 MB) and its input samples --
// which lanes share eight apart -- from LDS.  Per tap: one 8-byte load, one LDS read, a conversion, a multiplication, an
// addition; products of taps outside the file are w * (+0.0), which leaves the sum as it is (it is never -0.0).
__device__ __forceinline__ float rational_sample(const FileDesc& f, const float* __restrict__ in, uint64_t n, bool active,
                                                 uint64_t n_first, uint64_t n_last, float* s_in) {
    // the block's input range: from the first output's first possible tap to the last output's last one
    const long kbase = (long)((n_first * f.ph_p) / f.ph_q) + (long)f.ph_m_min;
    const long kend = (long)((n_last * f.ph_p) / f.ph_q) + (long)f.ph_m_min + (long)f.ph_m_span - 1;
    const bool stage_in = kend - kbase + 1 <= (long)kInMax;
    __syncthreads();                                          // s_in is reused by consecutive blocks
    if (stage_in) {
        for (long p = threadIdx.x; p <= kend - kbase; p += kThreads) {
            const long kk = kbase + p;
            s_in[p + (p >> 6)] = kk >= 0 && (uint64_t)kk < f.n_in ? in[(uint64_t)kk] : 0.0f;
        }
    }
    __syncthreads();
    if (!active) return 0.0f;
    const uint64_t np = n * f.ph_p, r = np % f.ph_q;
    const long ip = (long)(np / f.ph_q), m0 = (long)f.ph_first[r];
    const uint32_t cnt = f.ph_count[r];
    const double* w = f.ph_w + (size_t)(m0 - (long)f.ph_m_min) * f.ph_q + r;
    double acc = 0.0;
    if (stage_in) {
        uint32_t q = (uint32_t)(ip + m0 - kbase);
        for (uint32_t j = 0; j < cnt; ++j, w += f.ph_q, ++q) acc += *w * (double)s_in[q + (q >> 6)];
    } else {
        for (uint32_t j = 0; j < cnt; ++j, w += f.ph_q) {
            const long k = ip + m0 + (long)j;
            if (k >= 0 && (uint64_t)k < f.n_in) acc += *w * (double)in[(uint64_t)k];
        }
    }
    const double wsum = f.ph_wsum[r];
    return (float)(wsum != 0.0 ? acc / wsum : 0.0);
}

// every file of a batch in one launch: blockIdx.y = file, blockIdx.x walks its output samples; a file whose rate is
// the processing rate is copied
__global__ __launch_bounds__(kThreads) void resample_batch_kernel(const FileDesc* __restrict__ files, const float* __restrict__ decoded,
                                                                  int res, const double* __restrict__ table, uint64_t table_n,
                                                                  float* __restrict__ pcm) {
    __shared__ double s_rows[kTapGroup][kRowLen];
    __shared__ uint32_t s_lo[kTapGroup];
    __shared__ uint32_t s_stat[8];
    __shared__ float s_in[kInMax + kInMax / 64 + 1];
    const FileDesc f = files[blockIdx.y];
    const float* in = decoded + f.dec_off + f.first;
    float* out = pcm + f.out_off;
    for (uint64_t n0 = (uint64_t)blockIdx.x * kThreads; n0 < f.n_write; n0 += (uint64_t)gridDim.x * kThreads) {
        const uint64_t n = n0 + threadIdx.x;
        const bool active = n < f.n_write;
        float v = 0.0f;
        if (f.copy) {
            if (active) v = in[n];
        } else if (f.mode == 2) {
            if (active) v = linear_sample(in, f.n_in, f.ratio, n);
        } else if (f.ph_q) {
            const uint64_t n_last = n0 + kThreads - 1 < f.n_write ? n0 + kThreads - 1 : f.n_write - 1;
            v = rational_sample(f, in, n, active, n0, n_last, s_in);
        } else {
            // (a lane past the end works on the file's last sample, so that every lane reads inside the staged runs)
            v = sinc_sample_tiled(in, f.n_in, f.ratio, f.scale, f.half, res, table, table_n, active ? n : f.n_write - 1, active,
                                  s_rows, s_lo, s_stat, s_in);
        }
        if (active) out[n] = v;
    }
}

}  // namespace

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
