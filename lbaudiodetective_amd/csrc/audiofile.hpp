// audiofile.hpp -- uncompressed CAF / WAV reader (stand-in for ExtAudioFile)
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace lbad {
enum class AudioFileStatus { Ok, NotFound, Unsupported };
// Reads the whole file as mono float32 at the file's own sample rate.
AudioFileStatus read_audio_file(const char* path, std::vector<float>& mono, double& sample_rate);

// The same in two steps, for the device decoder (k_decode.hip): the file's bytes with the payload located and
// described, then the decode.  Decoded mono frame i of the result is frame first + i of the payload's
// total_frames (CAF 'pakt' priming / valid-frame trimming); count frames in all.
struct AudioPayload {
    enum Kind { None = 0, Ima4 = 1, Pcm = 2, WavU8 = 3 };
    Kind kind = None;
    std::vector<uint8_t> file;        // the whole file (empty when the caller supplied the bytes)
    const uint8_t* bytes = nullptr;   // the file's bytes: file.data() or the caller's buffer
    size_t off = 0, len = 0;          // payload bytes within it
    uint32_t channels = 0, bits = 0;
    bool is_float = false, little = false;
    int64_t valid_frames = -1;
    int32_t priming = 0;
    double sample_rate = 0.0;
    uint64_t total_frames = 0, first = 0, count = 0;
};
AudioFileStatus parse_audio_file(const char* path, AudioPayload& out);
// the same on a file already in memory (the batch path reads files straight into pinned memory); `data` must
// outlive `out`
AudioFileStatus parse_audio_bytes(const uint8_t* data, size_t size, AudioPayload& out);
bool decode_payload(const AudioPayload& payload, std::vector<float>& mono);
// Sample-rate conversion (documented stand-ins for Apple's converter): mode 0 long Kaiser sinc, 1 short
// sinc, 2 linear interpolation.  False for an unknown mode or a rate ratio outside [1/4096, 4096].
bool resample(const std::vector<float>& in, double rate_in, double rate_out, uint32_t mode, std::vector<float>& out);

// What resample() evaluates, for the device version of the same arithmetic (k_resample.hip): output sample n is
// taken at input position n * ratio; modes 0 / 1 sum the inputs within `half` samples of it, weighted by the
// kernel table (`table_res` entries per unit of |k - pos| / scale, linear interpolation) and normalised by the sum
// of the weights; mode 2 interpolates linearly.  `copy`: the rates are equal, the output is the input.
//
// Rational position (round 4, modes 0 / 1).  When both rates are whole numbers of Hz, rate_in / rate_out = p / q in lowest
// terms and q <= 16384, output n sits at input position n p / q EXACTLY: ip = (n p) div q, phase = (n p) mod q, frac =
// (double)phase / (double)q.  Its taps are the inputs k = ip + m for the integers m with ceil(frac - half) <= m <=
// floor(frac + half); the weight of tap m is the kernel table read at |(double)m - frac| * coord (same interpolation), a tap
// the table does not cover is skipped; weights and products are summed in ascending m, products only for taps inside the
// file; the sample is acc / wsum (0 when wsum is 0).  Everything but the samples depends on the phase alone, so the q
// phases' weights are built ONCE (PhaseTable: 1378 x 385 doubles for 44.1 kHz -> 5512 Hz) and a tap costs one load and
// one multiply-add instead of the coordinate, the truncation and the table interpolation (22 vector instructions on the
// device).  Until round 4 the position was the double product n * ratio, whose rounding differs from output to output;
// rate pairs that are not rational in this sense keep that definition.  The host function, the device kernel and the
// independent oracle (oracle/lbad_file_oracle.c) all follow this paragraph.
struct PhaseTable {
    uint64_t p = 0, q = 0;
    int32_t m_min = 0;                 // smallest first tap over the phases
    uint32_t m_span = 0;               // rows of `w`: taps m_min .. m_min + m_span - 1
    std::vector<int32_t> first;        // per phase: first covered tap m
    std::vector<uint32_t> count;       // per phase: covered taps (they are contiguous: the coordinate is monotone in |m - frac|)
    std::vector<double> wsum;          // per phase: sum of its weights, ascending m
    std::vector<double> w;             // weight of tap m of phase r at [(m - m_min) * q + r]
};

struct ResamplePlan {
    uint32_t mode = 0;
    bool copy = false;
    uint64_t n_out = 0;
    double ratio = 1.0, scale = 1.0, half = 0.0;
    int table_res = 0;
    const std::vector<double>* table = nullptr;   // process-lifetime storage
    const PhaseTable* phases = nullptr;           // rational position (process-lifetime storage), else nullptr
};
bool resample_plan(uint64_t n_in, double rate_in, double rate_out, uint32_t mode, ResamplePlan& plan);
}  // namespace lbad
