// k_file_tail.hip -- the end-of-file windows of upstream's file loop, tail mode "stale"
// (LBAudioDetective.m:262-290 with ExtAudioFileRead delivering partial reads).
//
// Upstream runs the FFT in place in the buffer it reads into (:275,351-355).  A read that comes
// back short overwrites only the first nRead floats; the rest still holds the PREVIOUS window's
// packed spectrum, the transform still spans all W floats, and nRead replaces the window size in
// the band arithmetic (:373,382-383,390-395).  Every such window therefore depends on its
// predecessor: the windows of one file are a chain, one workgroup walks it.  Nothing here is
// throughput work (a 9 s file has ~200 of these windows); the arithmetic and its order are those of
// the canonical transform (radix-2 DIT, nested fmaf butterflies, folded split pass) so the rows
// equal oracle/lbad_oracle.c:lbo_fingerprint_file_loop bit for bit.
#include "internal.hpp"

namespace lbad {
namespace {

constexpr int kThreads = 256;

// in-place canonical real FFT of s[0..W) (W = 2N floats) through zr/zi; result packed like
// vDSP_fft_zrip + ztoc back into s
__device__ void chain_fft(float* s, float* zr, float* zi, uint32_t N, uint32_t logN, const float* __restrict__ twr,
                          const float* __restrict__ twi) {
    const uint32_t W = 2 * N;
    for (uint32_t i = threadIdx.x; i < N; i += kThreads) {
        const uint32_t n = __brev(i) >> (32 - logN);
        zr[i] = s[2 * n];
        zi[i] = s[2 * n + 1];
    }
    __syncthreads();
    for (uint32_t m = 2; m <= N; m <<= 1) {
        const uint32_t h = m >> 1, tstep = W / m;
        for (uint32_t t = threadIdx.x; t < N / 2; t += kThreads) {
            const uint32_t j = t & (h - 1);
            const uint32_t a = (t - j) * 2 + j, b = a + h;
            const float ur = zr[a], ui = zi[a], vr = zr[b], vi = zi[b];
            if (j == 0) {
                zr[a] = __fadd_rn(ur, vr); zi[a] = __fadd_rn(ui, vi);
                zr[b] = __fsub_rn(ur, vr); zi[b] = __fsub_rn(ui, vi);
            } else if (4 * j == m) {
                zr[a] = __fadd_rn(ur, vi); zi[a] = __fsub_rn(ui, vr);
                zr[b] = __fsub_rn(ur, vi); zi[b] = __fadd_rn(ui, vr);
            } else {
                const float wr = twr[j * tstep], wi = twi[j * tstep];
                zr[a] = __fmaf_rn(wr, vr, __fmaf_rn(-wi, vi, ur));
                zi[a] = __fmaf_rn(wr, vi, __fmaf_rn(wi, vr, ui));
                zr[b] = __fmaf_rn(-wr, vr, __fmaf_rn(wi, vi, ur));
                zi[b] = __fmaf_rn(-wr, vi, __fmaf_rn(-wi, vr, ui));
            }
        }
        __syncthreads();
    }
    for (uint32_t k = threadIdx.x; k < N; k += kThreads) {
        if (k == 0) {
            const float sm = __fadd_rn(zr[0], zi[0]), df = __fsub_rn(zr[0], zi[0]);
            s[0] = __fadd_rn(sm, sm);
            s[1] = __fadd_rn(df, df);
        } else {
            const float ar = zr[k], ai = zi[k], br = zr[N - k], bi = zi[N - k];
            const float sr = __fadd_rn(ar, br), si = __fsub_rn(ai, bi);
            const float dr = __fsub_rn(ar, br), di = __fadd_rn(ai, bi);
            const float wr = twr[k], wi = twi[k];
            s[2 * k] = __fmaf_rn(wr, di, __fmaf_rn(wi, dr, sr));
            s[2 * k + 1] = __fmaf_rn(-wr, dr, __fmaf_rn(wi, di, si));
        }
    }
    __syncthreads();
}

// tbl: per chained window [n_read, lo[bands], hi[bands]]
__global__ __launch_bounds__(kThreads) void file_tail_kernel(const float* __restrict__ pcm, uint64_t n_client,
                                                             uint32_t hop, uint32_t W, uint32_t logN,
                                                             const float* __restrict__ tw, uint32_t bands,
                                                             const uint32_t* __restrict__ band_tbl,
                                                             uint64_t first_short, uint32_t n_tail,
                                                             const uint32_t* __restrict__ tbl,
                                                             float* __restrict__ frames) {
    extern __shared__ float smem[];
    const uint32_t N = W / 2;
    float* s = smem;
    float* zr = smem + W;
    float* zi = zr + N;
    const float* twr = tw;
    const float* twi = tw + N;
    // the buffer as the last full window left it (:351-355); zeros when there is none
    if (first_short > 0) {
        const float* src = pcm + (first_short - 1) * (uint64_t)hop;
        for (uint32_t i = threadIdx.x; i < W; i += kThreads) s[i] = src[i];
        __syncthreads();
        chain_fft(s, zr, zi, N, logN, twr, twi);
    } else {
        for (uint32_t i = threadIdx.x; i < W; i += kThreads) s[i] = 0.0f;
        __syncthreads();
    }
    for (uint32_t t = 0; t < n_tail; ++t) {
        const uint64_t win = first_short + t;
        const uint32_t* e = tbl + (size_t)t * (1 + 2 * bands);
        const uint32_t n_read = e[0];
        const float* src = pcm + win * (uint64_t)hop;
        for (uint32_t i = threadIdx.x; i < n_read; i += kThreads) s[i] = src[i];   // :275
        __syncthreads();
        chain_fft(s, zr, zi, N, logN, twr, twi);
        if (threadIdx.x < bands) {                                                 // :373-405 with nRead
            const uint32_t b = threadIdx.x;
            const uint32_t width = n_read / 2;
            const float norm = (float)(width / 2);
            float p = 0.0f;
            for (uint32_t k = e[1 + b]; k < e[1 + bands + b]; ++k) {
                float re = s[2 * k], im = s[2 * k + 1];
                if (re > 0.0f) re = __fdiv_rn(re, norm);
                if (im > 0.0f) im = __fdiv_rn(im, norm);
                const float v = __fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im));
                if (v == v && !isinf(v)) p = __fadd_rn(p, v);
            }
            frames[win * bands + b] = __fdiv_rn(p, __uint_as_float(band_tbl[2 * bands + b]));
        }
        __syncthreads();
    }
}

// Tail mode 1: a window whose read delivers nothing has inNumberFrames == 0, the band loops are empty and the row
// is 0 / divisor in every band (LBAudioDetective.m:382-404) -- +0, or NaN for a band whose two edge indices coincide.
__global__ __launch_bounds__(kThreads) void empty_rows_kernel(float* __restrict__ rows, uint64_t n_values, uint32_t bands,
                                                              const uint32_t* __restrict__ band_tbl) {
    const uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i < n_values) rows[i] = __fdiv_rn(0.0f, __uint_as_float(band_tbl[2 * bands + (uint32_t)(i % bands)]));
}

// the same for every file of a batch: blockIdx.y = file, blockIdx.x walks the values of its short rows
__global__ __launch_bounds__(kThreads) void empty_rows_batch_kernel(const FileDesc* __restrict__ files, float* __restrict__ frames,
                                                                    uint32_t bands, const uint32_t* __restrict__ band_tbl) {
    const FileDesc f = files[blockIdx.y];
    if (f.first_short >= f.rows) return;
    float* rows = frames + (f.row_begin + f.first_short) * bands;
    const uint64_t n_values = (f.rows - f.first_short) * bands;
    for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < n_values; i += (uint64_t)gridDim.x * kThreads)
        rows[i] = __fdiv_rn(0.0f, __uint_as_float(band_tbl[2 * bands + (uint32_t)(i % bands)]));
}

}  // namespace

hipError_t launch_empty_rows_batch(const Plan& p, const FileDesc* d_files, uint32_t n_files, uint64_t max_rows, float* d_frames,
                                   hipStream_t stream) {
    if (n_files == 0 || max_rows == 0) return hipSuccess;
    if (n_files > 65535u) return hipErrorInvalidValue;
    uint64_t bx = (max_rows * p.bands + kThreads - 1) / kThreads;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(empty_rows_batch_kernel, dim3((uint32_t)bx, n_files), dim3(kThreads), 0, stream, d_files, d_frames, p.bands,
                       p.d_bands);
    return hipGetLastError();
}

hipError_t launch_empty_rows(const Plan& p, float* d_rows, uint64_t n_rows, hipStream_t stream) {
    const uint64_t n = n_rows * p.bands;
    if (n == 0) return hipSuccess;
    const uint64_t blocks = (n + kThreads - 1) / kThreads;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(empty_rows_kernel, dim3((uint32_t)blocks), dim3(kThreads), 0, stream, d_rows, n, p.bands, p.d_bands);
    return hipGetLastError();
}

hipError_t launch_file_tail(const Plan& p, const float* d_pcm, uint64_t n_client, uint32_t hop, uint64_t first_short,
                            uint32_t n_tail, const uint32_t* d_tbl, float* d_frames, hipStream_t stream) {
    if (n_tail == 0) return hipSuccess;
    const size_t lds = (size_t)2 * p.window * sizeof(float);
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(&file_tail_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(file_tail_kernel, dim3(1), dim3(kThreads), lds, stream, d_pcm, n_client, hop, p.window,
                       p.log2w - 1, p.d_tw, p.bands, p.d_bands, first_short, n_tail, d_tbl, d_frames);
    return hipGetLastError();
}

}  // namespace lbad
