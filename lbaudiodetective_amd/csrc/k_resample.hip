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
// Outputs n and n + q have the SAME phase, hence the same weights, and their inputs lie exactly p samples apart.  A block
// takes a run of up to 256 positions inside the period of q outputs and kPeriods consecutive periods: a thread loads
// each of its ~385 weights ONCE (8 bytes, contiguous across the block where p mod q = 1: 44.1 kHz -> 5512 Hz) and uses it
// for kPeriods outputs, whose input samples -- which lanes share eight apart -- come from kPeriods staged ranges in LDS.
// (Round 4, first version: one output per thread, i.e. 3 KB of weights per output through L2 -- 4.9 GB per sixty files,
// the whole kernel time.)  Per tap and output: one LDS read, a conversion, a multiplication, an addition, in the tap
// order of resample(); products of taps outside the file are w * (+0.0), which leaves the sum as it is (never -0.0).
#ifndef LBAD_RS_PERIODS
#define LBAD_RS_PERIODS 3
#endif
constexpr int kPeriods = LBAD_RS_PERIODS;
// Round 6: one pad word per EIGHT samples (sample q at word q + (q >> 3)): lanes 8 samples apart are 9 words apart, 32 lanes
// on 32 banks as before -- but now the pad a thread meets while it walks its taps repeats every eight taps, so the LDS
// address of tap 8 m + r is (a per-thread constant for r) + 9 m: the step loop below has no address arithmetic per tap.
// With a pad per 32 samples every (tap, output) paid an add, a shift, a mask and an add -- 48 integer instructions beside
// the 36 double-precision ones of four taps x three outputs: as much time as the arithmetic itself.
constexpr int kInMaxR = 2816;                           // samples of a staged range (256 outputs at ratio 8.7 + 420 taps: 2640)
constexpr int kInStride = kInMaxR + kInMaxR / 8 + 1;    // 3169 words, as before (three ranges: 38 KB, four workgroups per CU)
static_assert(kInMaxR <= kInMax, "the rational path's ranges fit the block sized for the tiled path");

__device__ __forceinline__ void rational_file(const FileDesc& f, const float* __restrict__ in, float* __restrict__ out,
                                              float (*s_in)[kInStride]) {
    const uint64_t Q = f.ph_q, P = f.ph_p;
    const uint64_t chunks = (Q + kThreads - 1) / kThreads;
    const uint64_t periods = (f.n_write + Q - 1) / Q;
    const uint64_t groups = (periods + kPeriods - 1) / kPeriods;
    for (uint64_t slot = blockIdx.x; slot < chunks * groups; slot += gridDim.x) {
        const uint64_t c = slot % chunks, g = slot / chunks;
        const uint64_t t0 = c * kThreads;                                   // first position of the run inside the period
        const uint64_t lanes = Q - t0 < (uint64_t)kThreads ? Q - t0 : (uint64_t)kThreads;
        long kbase[kPeriods];
        bool staged[kPeriods], any[kPeriods];
        __syncthreads();                                                    // s_in is reused by consecutive slots
#pragma unroll
        for (int k = 0; k < kPeriods; ++k) {
            const uint64_t nf = (g * kPeriods + (uint64_t)k) * Q + t0;      // the run's first output in period k
            any[k] = nf < f.n_write;
            staged[k] = false;
            kbase[k] = 0;
            if (!any[k]) continue;                                          // (uniform)
            const uint64_t nl = nf + lanes - 1 < f.n_write ? nf + lanes - 1 : f.n_write - 1;
            kbase[k] = (long)((nf * P) / Q) + (long)f.ph_m_min;
            const long kend = (long)((nl * P) / Q) + (long)f.ph_m_min + (long)f.ph_m_span - 1;
            staged[k] = kend - kbase[k] + 1 <= (long)kInMaxR;
            if (staged[k]) {
                for (long p = threadIdx.x; p <= kend - kbase[k]; p += kThreads) {
                    const long kk = kbase[k] + p;
                    s_in[k][p + (p >> 3)] = kk >= 0 && (uint64_t)kk < f.n_in ? in[(uint64_t)kk] : 0.0f;
                }
            }
        }
        __syncthreads();
        if (threadIdx.x >= lanes) continue;
        const uint64_t n0 = g * kPeriods * Q + t0 + threadIdx.x;            // this thread's output in the group's first period
        if (n0 >= f.n_write) continue;
        const uint64_t np = n0 * P, r = np % Q;
        const long ip0 = (long)(np / Q), m0 = (long)f.ph_first[r];
        const uint32_t cnt = f.ph_count[r];
        // (a global-memory pointer, said so: as a generic one every weight was a flat load that also counts as an LDS access,
        // and the waits for the samples' LDS reads then wait for the weights as well)
        typedef const __attribute__((address_space(1))) double* gdouble_p;
        gdouble_p w = (gdouble_p)(f.ph_w + (size_t)(m0 - (long)f.ph_m_min) * Q + r);
        double acc[kPeriods];
        bool act[kPeriods];
        uint32_t q[kPeriods];
        bool all_staged = true;
#pragma unroll
        for (int k = 0; k < kPeriods; ++k) {
            acc[k] = 0.0;
            act[k] = n0 + (uint64_t)k * Q < f.n_write;
            // (an output behind the file's end walks the front of its buffer instead: read, never stored)
            q[k] = act[k] ? (uint32_t)(ip0 + (long)((uint64_t)k * P) + m0 - kbase[k]) : 0u;
            all_staged = all_staged && (staged[k] || !any[k]);
        }
        if (all_staged) {                                                   // (uniform) the usual case
            // tap j = 8 m + r of output k reads sample q[k] + j, i.e. word (q[k] + (q[k] >> 3)) + 9 m + r + ((q[k] & 7) + r >> 3):
            // eight per-thread pointers per output, all moved on by 9 words per eight taps; taps in ascending order, the three
            // outputs side by side -- the operations and their order are those of the loop this replaces
            const float* at[kPeriods][8];
#pragma unroll
            for (int k = 0; k < kPeriods; ++k)
#pragma unroll
                for (int r = 0; r < 8; ++r) at[k][r] = &s_in[k][q[k] + (q[k] >> 3) + (uint32_t)r + (((q[k] & 7u) + (uint32_t)r) >> 3)];
            uint32_t j = 0;
            for (; j + 8 <= cnt; j += 8) {
                // eight taps at a time: their weights and the 24 samples are all requested before the first product (left to
                // itself the scheduler, short of registers, waited for every load right behind it)
                double wv[8];
                float x[kPeriods][8];
#pragma unroll
                for (int r = 0; r < 8; ++r) wv[r] = w[(size_t)r * Q];
#pragma unroll
                for (int k = 0; k < kPeriods; ++k)
#pragma unroll
                    for (int r = 0; r < 8; ++r) x[k][r] = *at[k][r];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int k = 0; k < kPeriods; ++k) acc[k] += wv[r] * (double)x[k][r];
                w += 8 * Q;
#pragma unroll
                for (int k = 0; k < kPeriods; ++k)
#pragma unroll
                    for (int r = 0; r < 8; ++r) at[k][r] += 9;
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int k = 0; k < kPeriods; ++k) q[k] += j;
            for (; j < cnt; ++j, w += Q) {                                  // (the last cnt mod 16 taps)
                const double wv = *w;
#pragma unroll
                for (int k = 0; k < kPeriods; ++k) {
                    acc[k] += wv * (double)s_in[k][q[k] + (q[k] >> 3)];
                    ++q[k];
                }
            }
        } else {
            for (uint32_t j = 0; j < cnt; ++j, w += Q) {
                const double wv = *w;
#pragma unroll
                for (int k = 0; k < kPeriods; ++k) {
                    const long kk = ip0 + (long)((uint64_t)k * P) + m0 + (long)j;
                    if (act[k] && kk >= 0 && (uint64_t)kk < f.n_in) acc[k] += wv * (double)in[(uint64_t)kk];
                }
            }
        }
        const double wsum = f.ph_wsum[r];
#pragma unroll
        for (int k = 0; k < kPeriods; ++k)
            if (act[k]) out[n0 + (uint64_t)k * Q] = (float)(wsum != 0.0 ? acc[k] / wsum : 0.0);
    }
}

// every file of a batch in one launch: blockIdx.y = file, blockIdx.x walks its output samples; a file whose rate is
// the processing rate is copied
__global__ __launch_bounds__(kThreads) void resample_batch_kernel(const FileDesc* __restrict__ files, const float* __restrict__ decoded,
                                                                  int res, const double* __restrict__ table, uint64_t table_n,
                                                                  float* __restrict__ pcm) {
    // one block of LDS, two uses: the tiled path's table rows + one staged input range, or the rational path's kPeriods ranges
    constexpr size_t kTiledBytes = sizeof(double) * kTapGroup * kRowLen + sizeof(float) * (kInMax + kInMax / 64 + 1);
    constexpr size_t kRationalBytes = sizeof(float) * kPeriods * kInStride;
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[kTiledBytes > kRationalBytes ? kTiledBytes : kRationalBytes];
    __shared__ uint32_t s_lo[kTapGroup];
    __shared__ uint32_t s_stat[8];
    double (*s_rows)[kRowLen] = reinterpret_cast<double (*)[kRowLen]>(s_raw);
    float* s_in = reinterpret_cast<float*>(s_raw + sizeof(double) * kTapGroup * kRowLen);
    const FileDesc f = files[blockIdx.y];
    const float* in = decoded + f.dec_off + f.first;
    float* out = pcm + f.out_off;
    if (!f.copy && f.mode != 2 && f.ph_q) {
        rational_file(f, in, out, reinterpret_cast<float (*)[kInStride]>(s_raw));
        return;
    }
    for (uint64_t n0 = (uint64_t)blockIdx.x * kThreads; n0 < f.n_write; n0 += (uint64_t)gridDim.x * kThreads) {
        const uint64_t n = n0 + threadIdx.x;
        const bool active = n < f.n_write;
        float v = 0.0f;
        if (f.copy) {
            if (active) v = in[n];
        } else if (f.mode == 2) {
            if (active) v = linear_sample(in, f.n_in, f.ratio, n);
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
