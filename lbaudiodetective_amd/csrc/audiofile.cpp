// audiofile.cpp -- file front end standing in for ExtAudioFile (LBAudioDetective.m:224-237): CAF with
// 'lpcm' or 'ima4' (Apple IMA ADPCM, what the upstream bird fixtures use) payloads and RIFF/WAVE
// (PCM / IEEE float) -> mono float32 at the file's rate; resample() converts to the processing
// rate.  Host code, like the decoder it replaces; the fingerprint arithmetic stays on the GPU.
#include "audiofile.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <system_error>
#include <thread>

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

// ---- Apple IMA4: 34-byte packets = 2-byte big-endian header (9-bit predictor, 7-bit step index) +
//      64 4-bit codes, low nibble first (the published IMA/DVI ADPCM recurrence) ----------------------
const int kImaIndex[16] = {-1, -1, -1, -1, 2, 4, 6, 8, -1, -1, -1, -1, 2, 4, 6, 8};
const int kImaStep[89] = {7,     8,     9,     10,    11,    12,    13,    14,    16,    17,    19,    21,    23,
                          25,    28,    31,    34,    37,    41,    45,    50,    55,    60,    66,    73,    80,
                          88,    97,    107,   118,   130,   143,   157,   173,   190,   209,   230,   253,   279,
                          307,   337,   371,   408,   449,   494,   544,   598,   658,   724,   796,   876,   963,
                          1060,  1166,  1282,  1411,  1552,  1707,  1878,  2066,  2272,  2499,  2749,  3024,  3327,
                          3660,  4026,  4428,  4871,  5358,  5894,  6484,  7132,  7845,  8630,  9493,  10442, 11487,
                          12635, 13899, 15289, 16818, 18500, 20350, 22385, 24623, 27086, 29794, 32767};

bool decode_ima4(const uint8_t* data, size_t n_bytes, uint32_t channels, int64_t valid_frames, int32_t priming,
                 std::vector<float>& out) {
    if (channels == 0) return false;
    const size_t packets = n_bytes / (34 * (size_t)channels);   // packets are interleaved per channel
    std::vector<float> acc(packets * 64, 0.0f);
    for (size_t p = 0; p < packets; ++p) {
        for (uint32_t c = 0; c < channels; ++c) {
            const uint8_t* pk = data + (p * channels + c) * 34;
            const int header = (pk[0] << 8) | pk[1];
            int predictor = (int16_t)(header & 0xFF80);
            int index = header & 0x7F;
            if (index > 88) index = 88;
            for (int i = 0; i < 64; ++i) {
                const int nib = (i & 1) ? (pk[2 + (i >> 1)] >> 4) : (pk[2 + (i >> 1)] & 0x0F);
                const int step = kImaStep[index];
                int diff = step >> 3;
                if (nib & 4) diff += step;
                if (nib & 2) diff += step >> 1;
                if (nib & 1) diff += step >> 2;
                predictor += (nib & 8) ? -diff : diff;
                if (predictor > 32767) predictor = 32767;
                if (predictor < -32768) predictor = -32768;
                index += kImaIndex[nib];
                if (index < 0) index = 0;
                if (index > 88) index = 88;
                acc[p * 64 + i] += (float)predictor / 32768.0f;
            }
        }
    }
    if (channels > 1)
        for (float& v : acc) v /= (float)channels;
    size_t first = priming > 0 ? (size_t)priming : 0;
    if (first > acc.size()) first = acc.size();
    size_t count = acc.size() - first;
    if (valid_frames > 0 && (size_t)valid_frames < count) count = (size_t)valid_frames;   // 'pakt' trims the tail
    out.assign(acc.begin() + first, acc.begin() + first + count);
    return true;
}

}  // namespace

// Container parsing: which bytes of the file are the payload and how they decode.  The validity checks the
// decoders relied on happen here, so that the host and the device decoder (k_decode.hip) see the same accepted set.
AudioFileStatus parse_audio_file(const char* path, AudioPayload& out) {
    out = AudioPayload();
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
    if (got != buf.size()) return AudioFileStatus::Unsupported;
    const AudioFileStatus st = parse_audio_bytes(buf.data(), buf.size(), out);
    out.file = std::move(buf);                                // the vector's storage does not move with it
    out.bytes = out.file.data();
    return st;
}

AudioFileStatus parse_audio_bytes(const uint8_t* p, size_t n, AudioPayload& out) {
    out = AudioPayload();
    out.bytes = p;
    if (!p || n < 12) return AudioFileStatus::Unsupported;
    auto pcm_ok = [&]() {
        if (out.channels == 0 || out.bits == 0 || out.bits % 8 != 0 || out.bits > 64) return false;
        if (out.is_float && out.bits != 32 && out.bits != 64) return false;
        if (!out.is_float && out.bits != 8 && out.bits != 16 && out.bits != 24 && out.bits != 32) return false;
        return true;
    };
    auto finish_pcm = [&]() {
        const size_t frame = (size_t)out.channels * (out.bits / 8);
        out.total_frames = out.len / frame;
        out.first = 0;
        out.count = out.total_frames;
    };

    if (std::memcmp(p, "caff", 4) == 0) {
        size_t at = 8;
        bool have_desc = false, ima4 = false;
        while (at + 12 <= n) {
            const uint8_t* ch = p + at;
            const int64_t csz = (int64_t)be64(ch + 4);
            const size_t body = at + 12;
            size_t len = csz < 0 ? n - body : (size_t)csz;
            if (body + len > n) len = n - body;
            if (std::memcmp(ch, "desc", 4) == 0 && len >= 32) {
                out.sample_rate = be_f64(p + body);
                ima4 = std::memcmp(p + body + 8, "ima4", 4) == 0;
                if (!ima4 && std::memcmp(p + body + 8, "lpcm", 4) != 0) return AudioFileStatus::Unsupported;
                if (ima4 && (be32(p + body + 20) != 64 || be32(p + body + 24) == 0 || be32(p + body + 24) > 64 ||
                             be32(p + body + 16) != 34u * be32(p + body + 24)))
                    return AudioFileStatus::Unsupported;
                if (!(out.sample_rate > 0.0) || !std::isfinite(out.sample_rate)) return AudioFileStatus::Unsupported;
                const uint32_t flags = be32(p + body + 12);
                out.is_float = flags & 1u;
                out.little = flags & 2u;
                out.channels = be32(p + body + 24);
                out.bits = be32(p + body + 28);
                have_desc = true;
            } else if (std::memcmp(ch, "pakt", 4) == 0 && len >= 24) {
                out.valid_frames = (int64_t)be64(p + body + 8);     // mNumberValidFrames
                out.priming = (int32_t)be32(p + body + 16);          // mPrimingFrames
            } else if (std::memcmp(ch, "data", 4) == 0) {
                if (!have_desc || len < 4) return AudioFileStatus::Unsupported;
                out.off = body + 4;
                out.len = len - 4;
                if (ima4) {
                    if (out.channels == 0) return AudioFileStatus::Unsupported;
                    out.kind = AudioPayload::Ima4;
                    const size_t packets = out.len / (34 * (size_t)out.channels);   // packets are interleaved per channel
                    out.total_frames = (uint64_t)packets * 64;
                    uint64_t first = out.priming > 0 ? (uint64_t)out.priming : 0;
                    if (first > out.total_frames) first = out.total_frames;
                    uint64_t count = out.total_frames - first;
                    if (out.valid_frames > 0 && (uint64_t)out.valid_frames < count) count = (uint64_t)out.valid_frames;   // 'pakt' trims the tail
                    out.first = first;
                    out.count = count;
                    return AudioFileStatus::Ok;
                }
                if (!pcm_ok()) return AudioFileStatus::Unsupported;
                out.kind = AudioPayload::Pcm;
                finish_pcm();
                return AudioFileStatus::Ok;
            }
            at = body + len;
        }
        return AudioFileStatus::Unsupported;
    }

    if (std::memcmp(p, "RIFF", 4) == 0 && std::memcmp(p + 8, "WAVE", 4) == 0) {
        size_t at = 12;
        bool have_fmt = false;
        while (at + 8 <= n) {
            const uint8_t* ch = p + at;
            size_t len = le32(ch + 4);
            const size_t body = at + 8;
            if (body + len > n) len = n - body;
            if (std::memcmp(ch, "fmt ", 4) == 0 && len >= 16) {
                uint16_t tag = le16(p + body);
                out.channels = le16(p + body + 2);
                out.sample_rate = (double)le32(p + body + 4);
                if (!(out.sample_rate > 0.0)) return AudioFileStatus::Unsupported;
                out.bits = le16(p + body + 14);
                if (tag == 0xFFFE && len >= 26) tag = le16(p + body + 24);  // WAVE_FORMAT_EXTENSIBLE
                if (tag != 1 && tag != 3) return AudioFileStatus::Unsupported;
                out.is_float = tag == 3;
                out.little = true;
                have_fmt = true;
            } else if (std::memcmp(ch, "data", 4) == 0) {
                if (!have_fmt) return AudioFileStatus::Unsupported;
                out.off = body;
                out.len = len;
                if (!out.is_float && out.bits == 8) {                       // 8-bit WAV is unsigned
                    if (out.channels == 0) return AudioFileStatus::Unsupported;
                    out.kind = AudioPayload::WavU8;
                    finish_pcm();
                    return AudioFileStatus::Ok;
                }
                if (!pcm_ok()) return AudioFileStatus::Unsupported;
                out.kind = AudioPayload::Pcm;
                finish_pcm();
                return AudioFileStatus::Ok;
            }
            at = body + len + (len & 1);
        }
        return AudioFileStatus::Unsupported;
    }
    return AudioFileStatus::Unsupported;
}

bool decode_payload(const AudioPayload& a, std::vector<float>& mono) {
    const uint8_t* data = a.bytes + a.off;
    switch (a.kind) {
        case AudioPayload::Ima4: return decode_ima4(data, a.len, a.channels, a.valid_frames, a.priming, mono);
        case AudioPayload::Pcm: return decode_pcm(data, a.len, a.channels, a.bits, a.is_float, a.little, mono);
        case AudioPayload::WavU8: {
            mono.resize(a.count);
            for (size_t i = 0; i < a.count; ++i) {
                double acc = 0;
                for (uint32_t c = 0; c < a.channels; ++c) acc += ((int)data[i * a.channels + c] - 128) / 128.0;
                mono[i] = (float)(acc / a.channels);
            }
            return true;
        }
        default: return false;
    }
}

AudioFileStatus read_audio_file(const char* path, std::vector<float>& mono, double& sample_rate) {
    AudioPayload a;
    const AudioFileStatus st = parse_audio_file(path, a);
    if (st != AudioFileStatus::Ok) return st;
    sample_rate = a.sample_rate;
    return decode_payload(a, mono) ? AudioFileStatus::Ok : AudioFileStatus::Unsupported;
}


// ---- sample-rate conversion -------------------------------------------------------------------------
// Apple's converter is closed source; these are documented stand-ins.  Mode 0 (default): band-limited
// interpolation with a Kaiser-windowed sinc (beta 9, 24 zero crossings each side at the lower of the two
// rates, cut-off 0.92 of the lower Nyquist), evaluated in double precision at position n * rate_in /
// rate_out for output sample n and normalised to unit DC gain per output sample.  The kernel table is read at
// |k - pos| * c with c = 2048 / scale computed once (round 3: one multiplication per tap instead of a division
// and a multiplication -- a third of the device converter's time).  Mode 1: the same with a
// short kernel (4 zero crossings, beta 3, cut-off at the Nyquist frequency: a wide transition band that
// lets the octave above Nyquist alias in at -20..-40 dB).  Mode 2: linear interpolation between the two
// neighbouring input samples (no anti-alias filter at all).  HISTORY.md (rounds 1-3 text, section 8.1) measures the three against
// the essay's figures: the long kernel and the short one are indistinguishable there, linear
// interpolation is ruled out by Tests 3.1 / 3.2.
namespace {
double bessel_i0(double x) {
    double sum = 1.0, term = 1.0;
    const double q = x * x / 4.0;
    for (int k = 1; k < 64; ++k) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < 1e-17 * sum) break;
    }
    return sum;
}
}  // namespace

// The weights of every phase of a rational rate pair (audiofile.hpp), built once per (p, q, mode) and kept for the life of
// the process; nullptr when the pair is not rational in that sense (or the table would be unreasonably large).
static const PhaseTable* phase_table(double rate_in, double rate_out, const ResamplePlan& plan) {
    if (rate_in != std::floor(rate_in) || rate_out != std::floor(rate_out) || rate_in > 4294967295.0 || rate_out > 4294967295.0)
        return nullptr;
    uint64_t a = (uint64_t)rate_in, b = (uint64_t)rate_out;
    while (b) { const uint64_t t = a % b; a = b; b = t; }
    const uint64_t p = (uint64_t)rate_in / a, q = (uint64_t)rate_out / a;
    if (q > 16384 || p > (1ull << 24)) return nullptr;
    if ((2.0 * plan.half + 2.0) * (double)q * 8.0 > 64.0 * 1024.0 * 1024.0) return nullptr;
    struct Key { uint64_t p, q; uint32_t mode; bool operator<(const Key& o) const { return p != o.p ? p < o.p : q != o.q ? q < o.q : mode < o.mode; } };
    static std::mutex lock;
    static std::map<Key, std::unique_ptr<PhaseTable>> cache;
    std::lock_guard<std::mutex> g(lock);
    std::unique_ptr<PhaseTable>& slot = cache[Key{p, q, plan.mode}];
    if (slot) return slot.get();
    std::unique_ptr<PhaseTable> t(new PhaseTable);
    t->p = p; t->q = q;
    const std::vector<double>& table = *plan.table;
    const double coord = (double)plan.table_res / plan.scale, half = plan.half;
    t->first.resize(q); t->count.resize(q); t->wsum.resize(q);
    auto weight = [&](long m, double frac, double& w) -> bool {       // false: the table does not cover the tap
        const double x = std::fabs((double)m - frac) * coord;
        const size_t i = (size_t)x;
        if (i + 1 >= table.size()) return false;
        w = table[i] + (table[i + 1] - table[i]) * (x - (double)i);
        return true;
    };
    long m_min = 0, m_max = 0;
    bool any = false;
    for (uint64_t r = 0; r < q; ++r) {
        const double frac = (double)r / (double)q;
        const long c0 = (long)std::ceil(frac - half), c1 = (long)std::floor(frac + half);
        long fm = 0, n = 0;
        for (long m = c0; m <= c1; ++m) {
            double w;
            if (!weight(m, frac, w)) { if (n) break; continue; }     // (covered taps are contiguous)
            if (!n) fm = m;
            ++n;
        }
        t->first[r] = (int32_t)fm; t->count[r] = (uint32_t)n;
        if (n) {
            if (!any || fm < m_min) m_min = fm;
            if (!any || fm + n - 1 > m_max) m_max = fm + n - 1;
            any = true;
        }
    }
    t->m_min = (int32_t)m_min;
    t->m_span = any ? (uint32_t)(m_max - m_min + 1) : 0u;
    t->w.assign((size_t)t->m_span * q, 0.0);
    for (uint64_t r = 0; r < q; ++r) {
        const double frac = (double)r / (double)q;
        double sum = 0.0;
        for (uint32_t j = 0; j < t->count[r]; ++j) {
            const long m = (long)t->first[r] + j;
            double w = 0.0;
            (void)weight(m, frac, w);
            sum += w;
            t->w[(size_t)(m - m_min) * q + r] = w;
        }
        t->wsum[r] = sum;
    }
    slot = std::move(t);
    return slot.get();
}

bool resample_plan(uint64_t n_in, double rate_in, double rate_out, uint32_t mode, ResamplePlan& plan) {
    plan = ResamplePlan();
    if (mode > 2 || !(rate_in > 0.0) || !(rate_out > 0.0) || !std::isfinite(rate_in) || !std::isfinite(rate_out))
        return false;
    const double ratio = rate_in / rate_out;                 // input samples per output sample
    if (!(ratio >= 1.0 / 4096.0) || !(ratio <= 4096.0)) return false;
    plan.mode = mode;
    plan.ratio = ratio;
    if (n_in == 0) return true;
    if (rate_in == rate_out) {
        plan.copy = true;
        plan.n_out = n_in;
        return true;
    }
    plan.n_out = (uint64_t)((double)n_in / ratio);
    if (mode == 2) return true;
    plan.scale = ratio > 1.0 ? ratio : 1.0;                  // kernel is stretched when decimating
    const double cutoff = mode == 0 ? 0.92 : 1.0;
    const int zero_crossings = mode == 0 ? 24 : 4;
    const double beta = mode == 0 ? 9.0 : 3.0, i0b = bessel_i0(beta);
    // kernel sampled 2048 times per unit of t in [0, zero_crossings], read with linear interpolation; the two
    // tables are built once per process (4 ms for the long one)
    const int res = 2048;
    static std::vector<double> tables[2];
    static std::once_flag built[2];
    std::call_once(built[mode], [&] {
        std::vector<double>& tb = tables[mode];
        tb.resize((size_t)zero_crossings * res + 2);
        for (size_t i = 0; i < tb.size(); ++i) {
            const double t = (double)i / res;
            const double u = t / zero_crossings;
            const double win = u < 1.0 ? bessel_i0(beta * std::sqrt(1.0 - u * u)) / i0b : 0.0;
            const double a = M_PI * cutoff * t;
            tb[i] = cutoff * (a < 1e-12 ? 1.0 : std::sin(a) / a) * win;
        }
    });
    plan.table = &tables[mode];
    plan.table_res = res;
    plan.half = zero_crossings * plan.scale;                 // kernel half-width in input samples
    plan.phases = phase_table(rate_in, rate_out, plan);
    return true;
}

bool resample(const std::vector<float>& in, double rate_in, double rate_out, uint32_t mode, std::vector<float>& out) {
    out.clear();
    ResamplePlan plan;
    if (!resample_plan(in.size(), rate_in, rate_out, mode, plan)) return false;
    if (in.empty()) return true;
    if (plan.copy) { out = in; return true; }
    const double ratio = plan.ratio;
    const uint64_t n_out = plan.n_out;
    out.resize(n_out);
    if (mode == 2) {
        for (uint64_t n = 0; n < n_out; ++n) {
            const double pos = (double)n * ratio;
            const size_t k = (size_t)pos;
            const double f = pos - (double)k;
            const double a = k < in.size() ? (double)in[k] : 0.0, b = k + 1 < in.size() ? (double)in[k + 1] : 0.0;
            out[n] = (float)(a * (1.0 - f) + b * f);
        }
        return true;
    }
    const double scale = plan.scale;
    const int res = plan.table_res;
    const std::vector<double>& table = *plan.table;
    const double half = plan.half;
    const double coord = (double)res / scale;                // table points per input sample
    // every output sample is independent: long inputs are split over the host's cores (same arithmetic per
    // sample, so the result does not depend on the split)
    const PhaseTable* ph = plan.phases;
    auto span = [&](uint64_t n_begin, uint64_t n_end) {
        for (uint64_t n = n_begin; ph && n < n_end; ++n) {           // rational position: the phase's weights are ready
            const uint64_t np = n * ph->p, r = np % ph->q;
            const long ip = (long)(np / ph->q), m0 = ph->first[r];
            const double* w = ph->w.data() + (size_t)(m0 - ph->m_min) * ph->q + r;
            double acc = 0.0;
            for (uint32_t j = 0; j < ph->count[r]; ++j, w += ph->q) {
                const long k = ip + m0 + (long)j;
                if (k >= 0 && (size_t)k < in.size()) acc += *w * (double)in[(size_t)k];
            }
            out[n] = (float)(ph->wsum[r] != 0.0 ? acc / ph->wsum[r] : 0.0);
        }
        for (uint64_t n = n_begin; !ph && n < n_end; ++n) {
            const double pos = (double)n * ratio;
            const long k0 = (long)std::ceil(pos - half), k1 = (long)std::floor(pos + half);
            double acc = 0.0, wsum = 0.0;
            for (long k = k0; k <= k1; ++k) {
                const double t = std::fabs((double)k - pos) * coord;            // table coordinate
                const size_t i = (size_t)t;
                if (i + 1 >= table.size()) continue;
                const double w = table[i] + (table[i + 1] - table[i]) * (t - (double)i);
                wsum += w;
                if (k >= 0 && (size_t)k < in.size()) acc += w * (double)in[(size_t)k];
            }
            out[n] = (float)(wsum != 0.0 ? acc / wsum : 0.0);
        }
    };
    const double taps = (double)n_out * (2.0 * half + 1.0);
    unsigned workers = std::thread::hardware_concurrency();
    if (workers > 16) workers = 16;
    if (taps < 2e6 || workers < 2) {
        span(0, n_out);
        return true;
    }
    std::vector<std::thread> pool;
    const uint64_t per = (n_out + workers - 1) / workers;
    uint64_t next = per < n_out ? per : n_out;               // [0, next) is this thread's; first range not handed out
    try {
        for (unsigned w = 1; w < workers && next < n_out; ++w) {
            const uint64_t e = next + per < n_out ? next + per : n_out;
            pool.emplace_back(span, next, e);
            next = e;
        }
    } catch (const std::system_error&) {
        // no more threads to be had: this one does what is left
    }
    span(0, per < n_out ? per : n_out);
    if (next < n_out) span(next, n_out);
    for (std::thread& th : pool) th.join();
    return true;
}

}  // namespace lbad
