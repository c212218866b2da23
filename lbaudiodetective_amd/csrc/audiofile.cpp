// audiofile.cpp -- minimal file front end standing in for ExtAudioFile (LBAudioDetective.m:224-237):
// uncompressed CAF ('lpcm') and RIFF/WAVE (PCM / IEEE float) -> mono float32 at the file's rate.
// Compressed CAF payloads (the upstream bird fixtures are IMA4) and sample-rate conversion are
// not implemented yet; callers get kLBAudioDetectiveUnsupportedFile.
#include "audiofile.hpp"

#include <cstdio>
#include <cstring>

namespace lbad {
namespace {

uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }
uint64_t be64(const uint8_t* p) { return (uint64_t)be32(p) << 32 | be32(p + 4); }
uint32_t le32(const uint8_t* p) { return (uint32_t)p[3] << 24 | (uint32_t)p[2] << 16 | (uint32_t)p[1] << 8 | p[0]; }
uint16_t le16(const uint8_t* p) { return (uint16_t)(p[1] << 8 | p[0]); }

double be_f64(const uint8_t* p) {
    const uint64_t u = be64(p);
    double d;
    std::memcpy(&d, &u, 8);
    return d;
}

// one sample of `bits` width at p -> float in [-1, 1)
float sample_to_float(const uint8_t* p, uint32_t bits, bool is_float, bool little) {
    uint8_t b[8];
    const uint32_t bytes = bits / 8;
    for (uint32_t i = 0; i < bytes; ++i) b[i] = little ? p[i] : p[bytes - 1 - i];  // b is little-endian now
    if (is_float) {
        if (bits == 32) { float f; std::memcpy(&f, b, 4); return f; }
        double d; std::memcpy(&d, b, 8); return (float)d;
    }
    int32_t v = 0;
    switch (bits) {
        case 8: return (float)(int8_t)b[0] / 128.0f;
        case 16: v = (int16_t)(b[0] | b[1] << 8); return (float)v / 32768.0f;
        case 24: v = (int32_t)((uint32_t)b[0] << 8 | (uint32_t)b[1] << 16 | (uint32_t)b[2] << 24) >> 8; return (float)v / 8388608.0f;
        case 32: v = (int32_t)((uint32_t)b[0] | (uint32_t)b[1] << 8 | (uint32_t)b[2] << 16 | (uint32_t)b[3] << 24); return (float)((double)v / 2147483648.0);
        default: return 0.0f;
    }
}

bool decode_pcm(const uint8_t* data, size_t n_bytes, uint32_t channels, uint32_t bits, bool is_float, bool little,
                std::vector<float>& out) {
    if (channels == 0 || bits == 0 || bits % 8 != 0 || bits > 64) return false;
    if (is_float && bits != 32 && bits != 64) return false;
    if (!is_float && bits != 8 && bits != 16 && bits != 24 && bits != 32) return false;
    const size_t frame = (size_t)channels * (bits / 8);
    const size_t frames = n_bytes / frame;
    out.resize(frames);
    for (size_t i = 0; i < frames; ++i) {
        const uint8_t* p = data + i * frame;
        if (channels == 1) {
            out[i] = sample_to_float(p, bits, is_float, little);
        } else {  // average the channels (the client format upstream is mono, LBAudioDetective.m:125)
            double acc = 0.0;
            for (uint32_t c = 0; c < channels; ++c) acc += sample_to_float(p + c * (bits / 8), bits, is_float, little);
            out[i] = (float)(acc / channels);
        }
    }
    return true;
}

}  // namespace

AudioFileStatus read_audio_file(const char* path, std::vector<float>& mono, double& sample_rate) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return AudioFileStatus::NotFound;
    std::vector<uint8_t> buf;
    std::fseek(f, 0, SEEK_END);
    const long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (sz <= 0) { std::fclose(f); return AudioFileStatus::Unsupported; }
    buf.resize((size_t)sz);
    const size_t got = std::fread(buf.data(), 1, buf.size(), f);
    std::fclose(f);
    if (got != buf.size() || buf.size() < 12) return AudioFileStatus::Unsupported;
    const uint8_t* p = buf.data();
    const size_t n = buf.size();

    if (std::memcmp(p, "caff", 4) == 0) {
        size_t at = 8;
        bool have_desc = false, is_float = false, little = false;
        uint32_t channels = 0, bits = 0;
        while (at + 12 <= n) {
            const uint8_t* ch = p + at;
            const int64_t csz = (int64_t)be64(ch + 4);
            const size_t body = at + 12;
            size_t len = csz < 0 ? n - body : (size_t)csz;
            if (body + len > n) len = n - body;
            if (std::memcmp(ch, "desc", 4) == 0 && len >= 32) {
                sample_rate = be_f64(p + body);
                if (std::memcmp(p + body + 8, "lpcm", 4) != 0) return AudioFileStatus::Unsupported;
                const uint32_t flags = be32(p + body + 12);
                is_float = flags & 1u;
                little = flags & 2u;
                channels = be32(p + body + 24);
                bits = be32(p + body + 28);
                have_desc = true;
            } else if (std::memcmp(ch, "data", 4) == 0) {
                if (!have_desc || len < 4) return AudioFileStatus::Unsupported;
                return decode_pcm(p + body + 4, len - 4, channels, bits, is_float, little, mono)
                           ? AudioFileStatus::Ok : AudioFileStatus::Unsupported;
            }
            at = body + len;
        }
        return AudioFileStatus::Unsupported;
    }

    if (std::memcmp(p, "RIFF", 4) == 0 && std::memcmp(p + 8, "WAVE", 4) == 0) {
        size_t at = 12;
        bool have_fmt = false, is_float = false;
        uint32_t channels = 0, bits = 0;
        while (at + 8 <= n) {
            const uint8_t* ch = p + at;
            size_t len = le32(ch + 4);
            const size_t body = at + 8;
            if (body + len > n) len = n - body;
            if (std::memcmp(ch, "fmt ", 4) == 0 && len >= 16) {
                uint16_t tag = le16(p + body);
                channels = le16(p + body + 2);
                sample_rate = (double)le32(p + body + 4);
                bits = le16(p + body + 14);
                if (tag == 0xFFFE && len >= 26) tag = le16(p + body + 24);  // WAVE_FORMAT_EXTENSIBLE
                if (tag != 1 && tag != 3) return AudioFileStatus::Unsupported;
                is_float = tag == 3;
                have_fmt = true;
            } else if (std::memcmp(ch, "data", 4) == 0) {
                if (!have_fmt) return AudioFileStatus::Unsupported;
                // 8-bit WAV is unsigned
                if (!is_float && bits == 8) {
                    const size_t frames = len / channels;
                    mono.resize(frames);
                    for (size_t i = 0; i < frames; ++i) {
                        double acc = 0;
                        for (uint32_t c = 0; c < channels; ++c) acc += ((int)p[body + i * channels + c] - 128) / 128.0;
                        mono[i] = (float)(acc / channels);
                    }
                    return AudioFileStatus::Ok;
                }
                return decode_pcm(p + body, len, channels, bits, is_float, true, mono) ? AudioFileStatus::Ok
                                                                                       : AudioFileStatus::Unsupported;
            }
            at = body + len + (len & 1);
        }
        return AudioFileStatus::Unsupported;
    }
    return AudioFileStatus::Unsupported;
}

}  // namespace lbad
