// audiofile.hpp -- uncompressed CAF / WAV reader (stand-in for ExtAudioFile)
#pragma once
#include <cstdint>
#include <vector>

namespace lbad {
enum class AudioFileStatus { Ok, NotFound, Unsupported };
// Reads the whole file as mono float32 at the file's own sample rate.
AudioFileStatus read_audio_file(const char* path, std::vector<float>& mono, double& sample_rate);
// Band-limited sample-rate conversion (documented stand-in for Apple's converter).
void resample(const std::vector<float>& in, double rate_in, double rate_out, std::vector<float>& out);
}  // namespace lbad
