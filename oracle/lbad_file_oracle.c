/*
 * lbad_file_oracle.c -- CPU oracle of the FILE front end: container, payload decode and sample-rate conversion.
 *
 * TEST INFRASTRUCTURE ONLY (see lbad_oracle.h).  Written from the published formats and from the converter
 * definitions in lbaudiodetective_amd/csrc/audiofile.hpp and HISTORY.md (rounds 1-3 text, section 8), sharing no source with lbaudiodetective_amd/csrc/audiofile.cpp or the device
 * kernels (k_decode.hip, k_resample.hip) it checks.
 *
 * What it stands in for upstream: ExtAudioFileOpenURL / ExtAudioFileRead with a mono float32 client format at the
 * processing rate (LBAudioDetective/LBAudioDetective.m:224-237,275) -- Apple code that is closed source and absent
 * from the reference tree.  PARITY UNPINNED against Apple at three points, none of which the reference holds a
 * vector for:
 *   - Apple IMA4 decode is restated from the published IMA/DVI ADPCM recurrence as QuickTime packs it (34-byte
 *     packets: 16-bit big-endian header = predictor's top 9 bits + 7-bit step index, then 64 codes, low nibble
 *     first; the step is applied as step>>3 + the three conditional shifts).  Every packet restarts from its
 *     header; whether Apple's decoder instead carries the running predictor across packets when the header is
 *     within its 7-bit quantisation is not observable here.
 *   - the channel mix-down for a mono client format is taken to be the arithmetic mean;
 *   - the sample-rate converter is one of three documented models (below), not Apple's.
 * The essay's fifty published match percentages (tests/golden/essay_figures.json) pin the result end to end.
 *
 * Containers: CAF (Apple "Core Audio Format" spec: 'caff' header, chunks of 4-byte type + signed 64-bit big-endian
 * size; 'desc' = CAFAudioDescription, 'pakt' = packet table header, 'data' = 4-byte edit count + payload, size -1 =
 * to the end of the file) with 'lpcm' or 'ima4' payloads; RIFF/WAVE (little-endian chunks, 'fmt ' = WAVEFORMATEX
 * with tags 1 = PCM, 3 = IEEE float, 0xFFFE = extensible with the tag in the sub-format; 8-bit PCM is unsigned).
 */
#include "lbad_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- byte readers ------------------------------------------------------------------------------------ */
static uint64_t rd_be(const uint8_t* p, int n) {
    uint64_t v = 0;
    for (int i = 0; i < n; ++i) v = (v << 8) | p[i];
    return v;
}
static uint64_t rd_le(const uint8_t* p, int n) {
    uint64_t v = 0;
    for (int i = n - 1; i >= 0; --i) v = (v << 8) | p[i];
    return v;
}
static double as_f64(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static float as_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

enum { PAY_NONE = 0, PAY_IMA4 = 1, PAY_LPCM = 2, PAY_WAV_U8 = 3 };

typedef struct payload {
    int kind;
    const uint8_t* bytes;
    uint64_t n_bytes;
    uint32_t channels, bits;
    int is_float, little;
    int64_t valid;      /* 'pakt' mNumberValidFrames, -1 if absent */
    int32_t priming;    /* 'pakt' mPrimingFrames */
    double rate;
} payload;

static int lpcm_shape_ok(const payload* p) {
    if (p->channels == 0) return 0;
    if (p->is_float) return p->bits == 32 || p->bits == 64;
    return p->bits == 8 || p->bits == 16 || p->bits == 24 || p->bits == 32;
}

/* 0 = ok, 1 = not a supported file */
static int locate_payload(const uint8_t* f, uint64_t n, payload* out) {
    memset(out, 0, sizeof *out);
    out->valid = -1;
    if (n < 12) return 1;
    if (memcmp(f, "caff", 4) == 0) {
        int seen_desc = 0, is_ima = 0;
        uint64_t pos = 8;
        while (pos + 12 <= n) {
            const uint8_t* hd = f + pos;
            const int64_t declared = (int64_t)rd_be(hd + 4, 8);
            const uint64_t body = pos + 12;
            uint64_t len = declared < 0 ? n - body : (uint64_t)declared;
            if (len > n - body) len = n - body;
            if (memcmp(hd, "desc", 4) == 0 && len >= 32) {
                const uint8_t* d = f + body;
                out->rate = as_f64(rd_be(d, 8));
                const uint32_t flags = (uint32_t)rd_be(d + 12, 4), bytes_pp = (uint32_t)rd_be(d + 16, 4);
                const uint32_t frames_pp = (uint32_t)rd_be(d + 20, 4);
                out->channels = (uint32_t)rd_be(d + 24, 4);
                out->bits = (uint32_t)rd_be(d + 28, 4);
                if (memcmp(d + 8, "ima4", 4) == 0) {
                    is_ima = 1;
                    if (frames_pp != 64 || out->channels == 0 || out->channels > 64 || bytes_pp != 34u * out->channels) return 1;
                } else if (memcmp(d + 8, "lpcm", 4) == 0) {
                    is_ima = 0;
                } else {
                    return 1;
                }
                if (!(out->rate > 0.0) || !isfinite(out->rate)) return 1;
                out->is_float = (flags & 1u) != 0;   /* kCAFLinearPCMFormatFlagIsFloat */
                out->little = (flags & 2u) != 0;     /* kCAFLinearPCMFormatFlagIsLittleEndian */
                seen_desc = 1;
            } else if (memcmp(hd, "pakt", 4) == 0 && len >= 24) {
                out->valid = (int64_t)rd_be(f + body + 8, 8);
                out->priming = (int32_t)(uint32_t)rd_be(f + body + 16, 4);
            } else if (memcmp(hd, "data", 4) == 0) {
                if (!seen_desc || len < 4) return 1;
                out->bytes = f + body + 4;           /* after mEditCount */
                out->n_bytes = len - 4;
                if (is_ima) {
                    out->kind = PAY_IMA4;
                    return 0;
                }
                if (!lpcm_shape_ok(out)) return 1;
                out->kind = PAY_LPCM;
                return 0;
            }
            pos = body + len;
        }
        return 1;
    }
    if (memcmp(f, "RIFF", 4) == 0 && memcmp(f + 8, "WAVE", 4) == 0) {
        int seen_fmt = 0;
        uint64_t pos = 12;
        while (pos + 8 <= n) {
            const uint8_t* hd = f + pos;
            const uint64_t body = pos + 8;
            uint64_t len = rd_le(hd + 4, 4);
            if (len > n - body) len = n - body;
            if (memcmp(hd, "fmt ", 4) == 0 && len >= 16) {
                const uint8_t* d = f + body;
                uint32_t tag = (uint32_t)rd_le(d, 2);
                out->channels = (uint32_t)rd_le(d + 2, 2);
                out->rate = (double)rd_le(d + 4, 4);
                out->bits = (uint32_t)rd_le(d + 14, 2);
                if (tag == 0xFFFEu && len >= 26) tag = (uint32_t)rd_le(d + 24, 2);
                if (!(out->rate > 0.0) || (tag != 1 && tag != 3)) return 1;
                out->is_float = tag == 3;
                out->little = 1;
                seen_fmt = 1;
            } else if (memcmp(hd, "data", 4) == 0) {
                if (!seen_fmt) return 1;
                out->bytes = f + body;
                out->n_bytes = len;
                if (!out->is_float && out->bits == 8) {
                    if (out->channels == 0) return 1;
                    out->kind = PAY_WAV_U8;
                    return 0;
                }
                if (!lpcm_shape_ok(out)) return 1;
                out->kind = PAY_LPCM;
                return 0;
            }
            pos = body + len + (len & 1u);           /* chunks are word aligned */
        }
        return 1;
    }
    return 1;
}

/* ---- IMA ADPCM, QuickTime packing ---------------------------------------------------------------------- */
static const uint16_t ima_steps[89] = {
    7, 8, 9, 10, 11, 12, 13, 14, 16, 17, 19, 21, 23, 25, 28, 31, 34, 37, 41, 45, 50, 55, 60, 66, 73, 80, 88, 97, 107, 118,
    130, 143, 157, 173, 190, 209, 230, 253, 279, 307, 337, 371, 408, 449, 494, 544, 598, 658, 724, 796, 876, 963, 1060,
    1166, 1282, 1411, 1552, 1707, 1878, 2066, 2272, 2499, 2749, 3024, 3327, 3660, 4026, 4428, 4871, 5358, 5894, 6484,
    7132, 7845, 8630, 9493, 10442, 11487, 12635, 13899, 15289, 16818, 18500, 20350, 22385, 24623, 27086, 29794, 32767};

/* magnitude of the step a 3-bit code selects: step/8 + step/4 [b0] + step/2 [b1] + step [b2], each term truncated */
static int ima_magnitude(int step, int code3) {
    int m = step / 8;
    if (code3 & 1) m += step / 4;
    if (code3 & 2) m += step / 2;
    if (code3 & 4) m += step;
    return m;
}

/* How a packet starts: 0 (the model of this oracle and of the product) from its header; 1 (probe only,
 * tools/ima4_gap_probe.py) like decoders that carry the running predictor across packets whenever the header --
 * which stores only the predictor's top 9 bits -- agrees with it to within its quantisation step and the step index is
 * unchanged.  Which of the two Apple's decoder does cannot be observed here; the probe bounds what it could change. */
static int g_ima4_carry = 0;
void lbo_file_set_ima4_carry(int on) { g_ima4_carry = on; }

/* frames = packets * 64 mono floats (channels averaged) */
static void ima4_frames(const uint8_t* bytes, uint64_t packets, uint32_t channels, float* frames) {
    static const int8_t index_move[8] = {-1, -1, -1, -1, 2, 4, 6, 8};
    int run_sample[64], run_idx[64];
    for (int c = 0; c < 64; ++c) { run_sample[c] = 0; run_idx[c] = -1; }
    for (uint64_t pk = 0; pk < packets; ++pk) {
        float* dst = frames + pk * 64;
        for (uint32_t ch = 0; ch < channels; ++ch) {
            const uint8_t* p = bytes + (pk * channels + ch) * 34;
            const unsigned head = (unsigned)p[0] * 256u + p[1];
            int sample = (int)(head & 0xFF80u);
            if (sample >= 32768) sample -= 65536;     /* two's complement 16-bit */
            int idx = (int)(head & 0x7Fu);
            if (idx > 88) idx = 88;
            if (g_ima4_carry && ch < 64 && run_idx[ch] == idx && run_sample[ch] - sample > -128 && run_sample[ch] - sample < 128)
                sample = run_sample[ch];
            for (int k = 0; k < 64; ++k) {
                const unsigned byte = p[2 + k / 2];
                const unsigned code = (k % 2 == 0) ? (byte & 15u) : (byte >> 4);
                const int mag = ima_magnitude(ima_steps[idx], (int)(code & 7u));
                sample = (code & 8u) ? sample - mag : sample + mag;
                if (sample > 32767) sample = 32767;
                else if (sample < -32768) sample = -32768;
                idx += index_move[code & 7u];
                if (idx < 0) idx = 0;
                else if (idx > 88) idx = 88;
                const float v = (float)sample / 32768.0f;
                dst[k] = ch == 0 ? v : dst[k] + v;    /* float sum in channel order */
            }
            if (ch < 64) { run_sample[ch] = sample; run_idx[ch] = idx; }
        }
        if (channels > 1)
            for (int k = 0; k < 64; ++k) dst[k] = dst[k] / (float)channels;
    }
}

/* ---- linear PCM ---------------------------------------------------------------------------------------- */
static float lpcm_sample(const uint8_t* p, uint32_t bits, int is_float, int little) {
    const int n = (int)(bits / 8);
    const uint64_t raw = little ? rd_le(p, n) : rd_be(p, n);
    if (is_float) return bits == 32 ? as_f32((uint32_t)raw) : (float)as_f64(raw);
    /* sign-extend the n-byte integer, scale by 2^-(bits-1) */
    int64_t v = (int64_t)raw;
    if (v >> (bits - 1)) v -= (int64_t)1 << bits;
    if (bits == 32) return (float)((double)v / 2147483648.0);
    return (float)v / (float)((int64_t)1 << (bits - 1));
}

/* decode the payload to mono float at the file's rate; *out is malloc'ed.  0 = ok */
static int decode_payload_frames(const payload* p, float** out, uint64_t* n_out) {
    *out = NULL;
    *n_out = 0;
    if (p->kind == PAY_IMA4) {
        const uint64_t packets = p->n_bytes / (34ull * p->channels);
        const uint64_t total = packets * 64;
        float* all = (float*)malloc(sizeof(float) * (total ? total : 1));
        if (!all) return 2;
        ima4_frames(p->bytes, packets, p->channels, all);
        uint64_t skip = p->priming > 0 ? (uint64_t)p->priming : 0;
        if (skip > total) skip = total;
        uint64_t keep = total - skip;
        if (p->valid > 0 && (uint64_t)p->valid < keep) keep = (uint64_t)p->valid;   /* the packet table trims the tail */
        memmove(all, all + skip, sizeof(float) * keep);
        *out = all;
        *n_out = keep;
        return 0;
    }
    if (p->kind == PAY_LPCM || p->kind == PAY_WAV_U8) {
        const uint64_t frame_bytes = (uint64_t)p->channels * (p->bits / 8);
        const uint64_t frames = p->n_bytes / frame_bytes;
        float* all = (float*)malloc(sizeof(float) * (frames ? frames : 1));
        if (!all) return 2;
        for (uint64_t i = 0; i < frames; ++i) {
            const uint8_t* fr = p->bytes + i * frame_bytes;
            if (p->kind == PAY_WAV_U8) {
                double sum = 0.0;
                for (uint32_t c = 0; c < p->channels; ++c) sum += ((int)fr[c] - 128) / 128.0;
                all[i] = (float)(sum / p->channels);
            } else if (p->channels == 1) {
                all[i] = lpcm_sample(fr, p->bits, p->is_float, p->little);
            } else {
                double sum = 0.0;                      /* mean of the channels, in double */
                for (uint32_t c = 0; c < p->channels; ++c)
                    sum += lpcm_sample(fr + c * (p->bits / 8), p->bits, p->is_float, p->little);
                all[i] = (float)(sum / p->channels);
            }
        }
        *out = all;
        *n_out = frames;
        return 0;
    }
    return 1;
}

int lbo_file_decode_bytes(const uint8_t* file, uint64_t n_bytes, float** out_mono, uint64_t* out_frames, double* out_rate) {
    payload p;
    if (!file || locate_payload(file, n_bytes, &p) != 0) return 1;
    if (out_rate) *out_rate = p.rate;
    return decode_payload_frames(&p, out_mono, out_frames);
}

int lbo_file_decode(const char* path, float** out_mono, uint64_t* out_frames, double* out_rate) {
    FILE* f = fopen(path, "rb");
    if (!f) return -43;
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (sz <= 0) { fclose(f); return 1; }
    uint8_t* buf = (uint8_t*)malloc((size_t)sz);
    if (!buf) { fclose(f); return 2; }
    const size_t got = fread(buf, 1, (size_t)sz, f);
    fclose(f);
    int rc = got == (size_t)sz ? lbo_file_decode_bytes(buf, (uint64_t)sz, out_mono, out_frames, out_rate) : 1;
    free(buf);
    return rc;
}

void lbo_file_free(float* p) { free(p); }

/* ---- sample-rate conversion: the three documented models (audiofile.hpp; HISTORY.md section 8) -------------------------------
 * Output sample n is taken at input position x = n * r with r = rate_in / rate_out (double); the output has
 * floor(n_in / r) samples.
 *   model 2  linear interpolation between in[floor x] and in[floor x + 1] (0 past the end), in double.
 *   model 0 / 1  band-limited interpolation: weights w(k) = h(|k - x| / s) for every integer k with
 *     |k - x| <= Z s, s = max(r, 1); out = sum w(k) in[k] / sum w(k) (samples outside the input count as 0, their
 *     weights still enter the normalisation), all in double, k ascending.
 *     h(t) = c sinc(pi c t) I0(beta sqrt(1 - (t/Z)^2)) / I0(beta) for t < Z, 0 from Z on, read from a table of
 *     2048 points per unit of t with linear interpolation, at table coordinate |k - x| * (2048 / s) (the
 *     quotient formed once).
 *     model 0: Z = 24, beta = 9, c = 0.92.   model 1: Z = 4, beta = 3, c = 1.
 *   I0 by its power series, summed until a term falls below 1e-17 of the sum (at most 63 terms).
 * ---------------------------------------------------------------------------------------------------------- */
static double series_i0(double x) {
    const double q = x * x / 4.0;
    double term = 1.0, total = 1.0;
    for (int k = 1; k < 64; ++k) {
        term *= q / ((double)k * (double)k);
        total += term;
        if (term < 1e-17 * total) break;
    }
    return total;
}

#define SRC_POINTS_PER_UNIT 2048

static double* src_kernel_table(int model, size_t* n_points) {
    const int Z = model == 0 ? 24 : 4;
    const double beta = model == 0 ? 9.0 : 3.0, c = model == 0 ? 0.92 : 1.0;
    const double norm = series_i0(beta);
    const size_t n = (size_t)Z * SRC_POINTS_PER_UNIT + 2;
    double* tb = (double*)malloc(sizeof(double) * n);
    if (!tb) return NULL;
    for (size_t i = 0; i < n; ++i) {
        const double t = (double)i / SRC_POINTS_PER_UNIT;
        const double u = t / Z;
        const double window = u < 1.0 ? series_i0(beta * sqrt(1.0 - u * u)) / norm : 0.0;
        const double a = M_PI * c * t;
        tb[i] = c * (a < 1e-12 ? 1.0 : sin(a) / a) * window;
    }
    *n_points = n;
    return tb;
}

uint64_t lbo_resample_count(uint64_t n_in, double rate_in, double rate_out) {
    if (!(rate_in > 0.0) || !(rate_out > 0.0)) return 0;
    if (rate_in == rate_out) return n_in;
    return (uint64_t)((double)n_in / (rate_in / rate_out));
}

/* out holds lbo_resample_count() samples.  0 = ok */
int lbo_resample(const float* in, uint64_t n_in, double rate_in, double rate_out, int model, float* out) {
    if (model < 0 || model > 2 || !(rate_in > 0.0) || !(rate_out > 0.0)) return 1;
    const double r = rate_in / rate_out;
    if (!(r >= 1.0 / 4096.0) || !(r <= 4096.0)) return 1;
    if (n_in == 0) return 0;
    if (rate_in == rate_out) {
        memcpy(out, in, sizeof(float) * n_in);
        return 0;
    }
    const uint64_t n_out = (uint64_t)((double)n_in / r);
    if (model == 2) {
        for (uint64_t n = 0; n < n_out; ++n) {
            const double x = (double)n * r;
            const uint64_t k = (uint64_t)x;
            const double frac = x - (double)k;
            const double left = k < n_in ? (double)in[k] : 0.0, right = k + 1 < n_in ? (double)in[k + 1] : 0.0;
            out[n] = (float)(left * (1.0 - frac) + right * frac);
        }
        return 0;
    }
    size_t n_tb = 0;
    double* tb = src_kernel_table(model, &n_tb);
    if (!tb) return 2;
    const double s = r > 1.0 ? r : 1.0;
    const double reach = (model == 0 ? 24 : 4) * s;
    const double per_sample = (double)SRC_POINTS_PER_UNIT / s;   /* table points per input sample */
    /* Rational position (the converter's definition since round 4, csrc/audiofile.hpp): with whole-number rates whose
     * ratio is P / Q in lowest terms and Q <= 16384, output n sits at input position n P / Q exactly -- whole part W,
     * remainder R -- and everything but the samples depends on R alone.  Restated here sample by sample, without the
     * product's table of phases: the taps are W + m, m from ceil(R / Q - reach) to floor(R / Q + reach), weighted by the
     * kernel table at |m - R / Q| * per_sample, summed in ascending m. */
    uint64_t P = 0, Q = 0;
    if (rate_in == floor(rate_in) && rate_out == floor(rate_out) && rate_in <= 4294967295.0 && rate_out <= 4294967295.0) {
        uint64_t a = (uint64_t)rate_in, b = (uint64_t)rate_out;
        while (b) { const uint64_t t = a % b; a = b; b = t; }
        P = (uint64_t)rate_in / a; Q = (uint64_t)rate_out / a;
        if (Q > 16384 || P > (1ull << 24) || (2.0 * reach + 2.0) * (double)Q * 8.0 > 64.0 * 1024.0 * 1024.0) P = Q = 0;
    }
    if (Q) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
        for (int64_t n = 0; n < (int64_t)n_out; ++n) {
            const uint64_t np = (uint64_t)n * P;
            const long whole = (long)(np / Q);
            const double part = (double)(np % Q) / (double)Q;
            const long first = (long)ceil(part - reach), last = (long)floor(part + reach);
            double num = 0.0, den = 0.0;
            for (long m = first; m <= last; ++m) {
                const double t = fabs((double)m - part) * per_sample;
                const size_t i = (size_t)t;
                if (i + 1 >= n_tb) continue;
                const double w = tb[i] + (tb[i + 1] - tb[i]) * (t - (double)i);
                den += w;
                const long k = whole + m;
                if (k >= 0 && (uint64_t)k < n_in) num += w * (double)in[k];
            }
            out[n] = (float)(den != 0.0 ? num / den : 0.0);
        }
        free(tb);
        return 0;
    }
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t n = 0; n < (int64_t)n_out; ++n) {
        const double x = (double)n * r;
        const long first = (long)ceil(x - reach), last = (long)floor(x + reach);
        double num = 0.0, den = 0.0;
        for (long k = first; k <= last; ++k) {
            const double t = fabs((double)k - x) * per_sample;
            const size_t i = (size_t)t;
            if (i + 1 >= n_tb) continue;
            const double w = tb[i] + (tb[i + 1] - tb[i]) * (t - (double)i);
            den += w;
            if (k >= 0 && (uint64_t)k < n_in) num += w * (double)in[k];
        }
        out[n] = (float)(den != 0.0 ? num / den : 0.0);
    }
    free(tb);
    return 0;
}

/* A whole file as upstream's ProcessAudioURL walks it (LBAudioDetective.m:208-308): decode, convert to the
 * processing rate, then the window loop -- in FILE-frame bookkeeping (hop_mode 1, SURVEY Q17, with the end-of-file
 * treatment tail_mode) or in processing-rate samples (hop_mode 0).  out_bools: *out_count x cfg->subfp_len
 * Booleans, malloc'ed.  0 = ok, -43 file not found, 1 unsupported. */
int lbo_fingerprint_file(const char* path, const lbo_config* cfg, int hop_mode, int tail_mode, int resampler,
                         uint8_t** out_bools, uint64_t* out_count) {
    *out_bools = NULL;
    *out_count = 0;
    float* mono = NULL;
    uint64_t frames = 0;
    double rate = 0.0;
    int rc = lbo_file_decode(path, &mono, &frames, &rate);
    if (rc != 0) return rc;
    const uint64_t n_client = lbo_resample_count(frames, rate, cfg->sample_rate);
    float* client = (float*)malloc(sizeof(float) * (n_client ? n_client : 1));
    if (!client) { free(mono); return 2; }
    rc = lbo_resample(mono, frames, rate, cfg->sample_rate, resampler, client);
    free(mono);
    if (rc != 0) { free(client); return rc; }
    uint64_t count;
    uint8_t* bools;
    if (hop_mode == 0) {
        count = lbo_subfingerprint_count(n_client, cfg->window, cfg->stride);
        bools = (uint8_t*)calloc((size_t)(count ? count : 1) * cfg->subfp_len, 1);
        if (count && lbo_fingerprint_pcm(client, n_client, cfg, bools) != count) rc = 1;
    } else {
        count = frames >= cfg->window ? ((frames - cfg->window) / cfg->stride) / LBO_ROWS_PER_FRAME : 0;
        double h = floor((double)cfg->stride * cfg->sample_rate / rate + 0.5);
        if (h < 1.0) h = 1.0;
        bools = (uint8_t*)calloc((size_t)(count ? count : 1) * cfg->subfp_len, 1);
        if (count && lbo_fingerprint_file_loop(client, n_client, frames, (uint32_t)h, tail_mode, cfg, bools, NULL, NULL) != count)
            rc = 1;
    }
    free(client);
    if (rc != 0) { free(bools); return rc; }
    *out_bools = bools;
    *out_count = count;
    return 0;
}

/* n files, one per thread (nthreads OpenMP threads, files handed out dynamically; the converter's own parallel loops run
 * on one thread inside: nested regions are off) -- the CPU baseline of a catalogue build.  out_bools[i] / out_counts[i] as
 * lbo_fingerprint_file's; returns the first non-zero status. */
int lbo_fingerprint_files(const char* const* paths, uint64_t n, const lbo_config* cfg, int hop_mode, int tail_mode, int resampler,
                          int nthreads, uint8_t** out_bools, uint64_t* out_counts) {
    int first = 0;
    if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
#endif
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        const int rc = lbo_fingerprint_file(paths[i], cfg, hop_mode, tail_mode, resampler, &out_bools[i], &out_counts[i]);
        if (rc != 0) {
#ifdef _OPENMP
#pragma omp critical
#endif
            { if (first == 0) first = rc; }
        }
    }
    return first;
}

