// k_decode.hip -- payload decoders of the file front end on the device: Apple IMA4 packets and linear PCM -> mono float32.
//
// What ExtAudioFile does for upstream before the first window is read (LBAudioDetective.m:224-237).  audiofile.cpp
// holds the host versions (decode_ima4 / decode_pcm / the unsigned 8-bit WAV loop); these are the same operations in
// the same order per output frame, so a file decoded here has the host decoder's samples bit for bit: IMA4 is integer
// arithmetic followed by one float division by 32768 per channel and a float sum over the channels in channel order;
// PCM is one conversion per sample and a double sum over the channels divided by the channel count.
#include "internal.hpp"

namespace lbad {
namespace {

constexpr int kThreads = 256;

__device__ const int kImaStep[89] = {7,     8,     9,     10,    11,    12,    13,    14,    16,    17,    19,    21,    23,
                                     25,    28,    31,    34,    37,    41,    45,    50,    55,    60,    66,    73,    80,
                                     88,    97,    107,   118,   130,   143,   157,   173,   190,   209,   230,   253,   279,
                                     307,   337,   371,   408,   449,   494,   544,   598,   658,   724,   796,   876,   963,
                                     1060,  1166,  1282,  1411,  1552,  1707,  1878,  2066,  2272,  2499,  2749,  3024,  3327,
                                     3660,  4026,  4428,  4871,  5358,  5894,  6484,  7132,  7845,  8630,  9493,  10442, 11487,
                                     12635, 13899, 15289, 16818, 18500, 20350, 22385, 24623, 27086, 29794, 32767};

// one thread per packet position: its `channels` interleaved 34-byte packets (2-byte big-endian header = 9-bit
// predictor + 7-bit step index, 64 4-bit codes, low nibble first), accumulated into the 64 output frames channel by
// channel as the host loop does (0.0f + v == v, so the first channel may store).  Round 4: the packet is read as
// seventeen 16-bit words (bytes when the payload starts at an odd offset), the step table comes from LDS, the index adjustment of a
// code is arithmetic (-1 for codes 0..3, 2 (code & 3) + 2 above), and the frames leave as sixteen 16-byte stores -- whole
// 64-byte lines per lane -- instead of 64 single floats (the kernel wrote 2.6 x its output in partial lines).
__device__ __forceinline__ void ima4_packet(const uint8_t* __restrict__ data, uint64_t p, uint32_t channels,
                                            float* __restrict__ out, const int* s_step) {
    float4* o = reinterpret_cast<float4*>(out + p * 64);         // (decoded offsets are 256-byte aligned)
    for (uint32_t c = 0; c < channels; ++c) {
        const uint8_t* pk8 = data + (p * channels + c) * 34;
        const bool even = (reinterpret_cast<uintptr_t>(pk8) & 1u) == 0;            // (a payload may start at an odd file offset)
        const uint16_t* pk16 = reinterpret_cast<const uint16_t*>(pk8);
        auto word_at = [&](int k) -> uint32_t {
            return even ? (uint32_t)pk16[k] : ((uint32_t)pk8[2 * k] | ((uint32_t)pk8[2 * k + 1] << 8));
        };
        const uint32_t h = word_at(0);
        const int header = (int)(((h & 0xFFu) << 8) | (h >> 8));   // big-endian on a little-endian load
        int predictor = (int)(short)(header & 0xFF80);
        int index = header & 0x7F;
        if (index > 88) index = 88;
#pragma unroll 2
        for (int g = 0; g < 16; ++g) {                            // four codes = one 16-bit word, low nibble of the low byte first
            const uint32_t word = word_at(1 + g);
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int nib = (int)((word >> (4 * i)) & 0xFu);
                const int step = s_step[index];
                int diff = step >> 3;
                if (nib & 4) diff += step;
                if (nib & 2) diff += step >> 1;
                if (nib & 1) diff += step >> 2;
                predictor += (nib & 8) ? -diff : diff;
                if (predictor > 32767) predictor = 32767;
                if (predictor < -32768) predictor = -32768;
                index += (nib & 4) ? 2 * (nib & 3) + 2 : -1;
                if (index < 0) index = 0;
                if (index > 88) index = 88;
                v[i] = __fdiv_rn((float)predictor, 32768.0f);
            }
            if (c == 0) {
                o[g] = float4{v[0], v[1], v[2], v[3]};
            } else {
                const float4 was = o[g];
                o[g] = float4{__fadd_rn(was.x, v[0]), __fadd_rn(was.y, v[1]), __fadd_rn(was.z, v[2]), __fadd_rn(was.w, v[3])};
            }
        }
    }
    if (channels > 1) {
        const float n = (float)channels;
        for (int g = 0; g < 16; ++g) {
            const float4 was = o[g];
            o[g] = float4{__fdiv_rn(was.x, n), __fdiv_rn(was.y, n), __fdiv_rn(was.z, n), __fdiv_rn(was.w, n)};
        }
    }
}

// one sample of `bits` width at p -> float in [-1, 1)  (audiofile.cpp: sample_to_float)
__device__ __forceinline__ float sample_to_float(const uint8_t* p, uint32_t bits, bool is_float, bool little) {
    uint8_t b[8];
    const uint32_t bytes = bits / 8;
    for (uint32_t i = 0; i < bytes; ++i) b[i] = little ? p[i] : p[bytes - 1 - i];  // b is little-endian now
    if (is_float) {
        if (bits == 32) return __uint_as_float((uint32_t)b[0] | (uint32_t)b[1] << 8 | (uint32_t)b[2] << 16 | (uint32_t)b[3] << 24);
        unsigned long long u = 0;
        for (int i = 0; i < 8; ++i) u |= (unsigned long long)b[i] << (8 * i);
        return (float)__longlong_as_double((long long)u);
    }
    int v = 0;
    switch (bits) {
        case 8: return __fdiv_rn((float)(signed char)b[0], 128.0f);
        case 16: v = (short)(b[0] | b[1] << 8); return __fdiv_rn((float)v, 32768.0f);
        case 24: v = (int)((uint32_t)b[0] << 8 | (uint32_t)b[1] << 16 | (uint32_t)b[2] << 24) >> 8; return __fdiv_rn((float)v, 8388608.0f);
        case 32: v = (int)((uint32_t)b[0] | (uint32_t)b[1] << 8 | (uint32_t)b[2] << 16 | (uint32_t)b[3] << 24); return (float)((double)v / 2147483648.0);
        default: return 0.0f;
    }
}

// one PCM frame; `wav_u8`: the unsigned 8-bit WAV form ((s - 128) / 128.0 in double, always averaged)
__device__ __forceinline__ void pcm_frame(const uint8_t* __restrict__ data, uint64_t i, uint32_t channels, uint32_t bits,
                                          int is_float, int little, int wav_u8, float* __restrict__ out) {
    if (wav_u8) {
        double acc = 0;
        for (uint32_t c = 0; c < channels; ++c) acc += ((int)data[i * channels + c] - 128) / 128.0;
        out[i] = (float)(acc / channels);
        return;
    }
    const uint32_t bytes = bits / 8;
    const uint8_t* p = data + i * ((uint64_t)channels * bytes);
    if (channels == 1) {
        out[i] = sample_to_float(p, bits, is_float, little);
    } else {  // average the channels (the client format upstream is mono, LBAudioDetective.m:125)
        double acc = 0.0;
        for (uint32_t c = 0; c < channels; ++c) acc += sample_to_float(p + c * bytes, bits, is_float, little);
        out[i] = (float)(acc / channels);
    }
}

// every file of a batch in one launch: blockIdx.y = file, blockIdx.x walks its packets (IMA4) or frames (PCM)
__global__ __launch_bounds__(kThreads) void decode_batch_kernel(const FileDesc* __restrict__ files, const uint8_t* __restrict__ bytes,
                                                                float* __restrict__ decoded) {
    __shared__ int s_step[89];
    if (threadIdx.x < 89) s_step[threadIdx.x] = kImaStep[threadIdx.x];
    __syncthreads();
    const FileDesc f = files[blockIdx.y];
    const uint64_t units = f.kind == 1 ? f.total_frames / 64 : f.total_frames;
    const uint8_t* data = bytes + f.bytes_off;
    float* out = decoded + f.dec_off;
    for (uint64_t u = (uint64_t)blockIdx.x * kThreads + threadIdx.x; u < units; u += (uint64_t)gridDim.x * kThreads) {
        if (f.kind == 1) ima4_packet(data, u, f.channels, out, s_step);
        else pcm_frame(data, u, f.channels, f.bits, (int)(f.flags & 1u), (int)((f.flags >> 1) & 1u), f.kind == 3 ? 1 : 0, out);
    }
}

}  // namespace

// d_files: n_files descriptors on the device; max_units: the largest packet / frame count among them
hipError_t launch_decode_batch(const FileDesc* d_files, uint32_t n_files, uint64_t max_units, const uint8_t* d_bytes,
                               float* d_decoded, hipStream_t stream) {
    if (n_files == 0 || max_units == 0) return hipSuccess;
    if (n_files > 65535u) return hipErrorInvalidValue;
    uint64_t bx = (max_units + kThreads - 1) / kThreads;
    if (bx > 4096) bx = 4096;
    hipLaunchKernelGGL(decode_batch_kernel, dim3((uint32_t)bx, n_files), dim3(kThreads), 0, stream, d_files, d_bytes, d_decoded);
    return hipGetLastError();
}

}  // namespace lbad
