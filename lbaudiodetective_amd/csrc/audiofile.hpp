// audiofile.hpp -- uncompressed CAF / WAV reader (stand-in for ExtAudioFile)
#pragma once
#include <cstdint>
#include <vector>

namespace lbad {
enum class AudioFileStatus { Ok, NotFound, Unsupported };
// Reads the whole file as mono float32 at the file's own sample rate.
AudioFileStatus read_audio_file(const char* path, std::vector<float>& mono, double& sample_rate);
// Sample-rate conversion (documented stand-ins for Apple's converter): mode 0 long Kaiser sinc, 1 short
// sinc, 2 linear interpolation.  False for an unknown mode or a rate ratio outside [1/4096, 4096].
bool resample(const std::vector<float>& in, double rate_in, double rate_out, uint32_t mode, std::vector<float>& out);
}  // namespace lbad
