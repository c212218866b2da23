/*
 * lbad_oracle.c -- CPU parity oracle (TEST INFRASTRUCTURE, see lbad_oracle.h).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; every fused multiply-add in the
 * canonical FFT is an explicit fmaf(), nothing else may be contracted).
 */
#include "lbad_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * configuration defaults -- LBAudioDetective.m:22-26,128
 * ---------------------------------------------------------------------------------------- */
void lbo_default_config(lbo_config* cfg) {
    cfg->sample_rate = 5512.0;
    cfg->window = 2048;
    cfg->stride = 64;
    cfg->bands = 32;
    cfg->subfp_len = 200;
}

static int is_pow2(uint32_t v) { return v && !(v & (v - 1)); }

static uint32_t ilog2(uint32_t v) {
    uint32_t l = 0;
    while ((1u << l) < v) ++l;
    return l;
}

/* ------------------------------------------------------------------------------------------
 * canonical real FFT
 *
 * The reference hands the window to vDSP_ctoz / vDSP_fft_zrip(FFT_RADIX2, FFT_FORWARD) /
 * vDSP_ztoc (LBAudioDetective.m:353-355).  vDSP is closed source; its documented contract is:
 * treat the W reals as W/2 complex values z[n] = x[2n] + i x[2n+1], transform, and return
 * 2 x DFT with the (real) DC and Nyquist terms packed into element 0.  This restatement
 * fixes ONE float32 evaluation order for that contract so that a GPU kernel can match it
 * bit for bit:
 *   - radix-2 decimation in time over the N = W/2 complex points, bit-reversed load,
 *     stages m = 2, 4, ..., N, butterflies in natural order;
 *   - twiddle j of stage m is tw[j * (W/m)] from the master table below;
 *   - j == 0 and j == m/4 butterflies are multiplication free; every other butterfly is
 *     out0 = u + w v, out1 = u - w v evaluated as two nested fmaf() per component
 *     (see bfly_general), one rounding per fmaf;
 *   - the split (real) pass and the x2 scaling are folded: Y[k] = (A + conj B) - i w (A - conj B)
 *     with A = Z[k], B = Z[N-k], w = tw[k], again as nested fmaf().
 * ---------------------------------------------------------------------------------------- */
void lbo_twiddles(uint32_t W, float* re, float* im) {
    const uint32_t half = W / 2, quarter = W / 4, eighth = W / 8;
    for (uint32_t k = 0; k < half; ++k) {
        /* fold k into the first octant [0, W/8] */
        uint32_t kk = k;
        int neg_cos = 0, swap = 0;
        if (kk > quarter) { kk = half - kk; neg_cos = 1; }   /* cos(pi - t) = -cos t, sin same */
        if (kk > eighth) { kk = quarter - kk; swap = 1; }    /* cos(pi/2 - t) = sin t          */
        const double ang = (2.0 * M_PI * (double)kk) / (double)W;
        float c = (float)cos(ang), s = (float)sin(ang);
        if (kk == eighth && W >= 8) s = c;                   /* cos(pi/4) == sin(pi/4) exactly */
        if (swap) { float t = c; c = s; s = t; }
        if (neg_cos) c = -c;
        if (k == quarter) c = 0.0f;                          /* cos(pi/2) */
        re[k] = c;
        im[k] = -s;
    }
}

static inline void bfly_general(float wr, float wi, float* ur, float* ui, float* vr, float* vi) {
    const float a = *ur, b = *ui, c = *vr, d = *vi;
    *ur = fmaf(wr, c, fmaf(-wi, d, a));
    *ui = fmaf(wr, d, fmaf(wi, c, b));
    *vr = fmaf(-wr, c, fmaf(wi, d, a));
    *vi = fmaf(-wr, d, fmaf(-wi, c, b));
}

typedef struct fft_plan {
    uint32_t W, N, logN;
    float* twr;
    float* twi;
    uint32_t* rev;
} fft_plan;

static int plan_init(fft_plan* p, uint32_t W) {
    if (!is_pow2(W) || W < 8) return -1;
    p->W = W;
    p->N = W / 2;
    p->logN = ilog2(p->N);
    p->twr = (float*)malloc(sizeof(float) * (W / 2));
    p->twi = (float*)malloc(sizeof(float) * (W / 2));
    p->rev = (uint32_t*)malloc(sizeof(uint32_t) * p->N);
    lbo_twiddles(W, p->twr, p->twi);
    for (uint32_t i = 0; i < p->N; ++i) {
        uint32_t r = 0;
        for (uint32_t b = 0; b < p->logN; ++b)
            if (i & (1u << b)) r |= 1u << (p->logN - 1 - b);
        p->rev[i] = r;
    }
    return 0;
}

static void plan_free(fft_plan* p) {
    free(p->twr);
    free(p->twi);
    free(p->rev);
}

/* zr/zi: scratch of N floats each.  The fma clone lets gcc inline fmaf() as one vfmadd where the
 * host has it; the default clone calls libm's (equally exact) fmaf. */
__attribute__((target_clones("fma", "default")))
static void rfft_exec(const fft_plan* p, const float* x, float* zr, float* zi, float* out) {
    const uint32_t N = p->N, W = p->W;
    for (uint32_t i = 0; i < N; ++i) {
        const uint32_t n = p->rev[i];
        zr[i] = x[2 * n];
        zi[i] = x[2 * n + 1];
    }
    for (uint32_t m = 2; m <= N; m <<= 1) {
        const uint32_t h = m >> 1, tstep = W / m;
        for (uint32_t base = 0; base < N; base += m) {
            for (uint32_t j = 0; j < h; ++j) {
                const uint32_t a = base + j, b = a + h;
                if (j == 0) {
                    const float ur = zr[a], ui = zi[a], vr = zr[b], vi = zi[b];
                    zr[a] = ur + vr; zi[a] = ui + vi;
                    zr[b] = ur - vr; zi[b] = ui - vi;
                } else if (4 * j == m) { /* w = -i */
                    const float ur = zr[a], ui = zi[a], vr = zr[b], vi = zi[b];
                    zr[a] = ur + vi; zi[a] = ui - vr;
                    zr[b] = ur - vi; zi[b] = ui + vr;
                } else {
                    bfly_general(p->twr[j * tstep], p->twi[j * tstep], &zr[a], &zi[a], &zr[b], &zi[b]);
                }
            }
        }
    }
    /* split pass, x2 scaling folded in */
    {
        const float s = zr[0] + zi[0], d = zr[0] - zi[0];
        out[0] = s + s;
        out[1] = d + d;
    }
    for (uint32_t k = 1; k < N; ++k) {
        const float ar = zr[k], ai = zi[k], br = zr[N - k], bi = zi[N - k];
        const float sr = ar + br, si = ai - bi;
        const float dr = ar - br, di = ai + bi;
        const float wr = p->twr[k], wi = p->twi[k];
        out[2 * k] = fmaf(wr, di, fmaf(wi, dr, sr));
        out[2 * k + 1] = fmaf(-wr, dr, fmaf(wi, di, si));
    }
}

int lbo_rfft_packed(const float* x, uint32_t W, float* out) {
    fft_plan p;
    if (plan_init(&p, W)) return -1;
    float* z = (float*)malloc(sizeof(float) * W);
    rfft_exec(&p, x, z, z + p.N, out);
    free(z);
    plan_free(&p);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * band edges and band means -- LBAudioDetective.m:361-405
 * ---------------------------------------------------------------------------------------- */
static uint32_t trunc_u32(double v) {
    /* (UInt32) of a double; negative values give 0 (what the ARM targets of the reference
     * do; C leaves values <= -1 undefined). */
    if (!(v > 0.0)) return 0;
    if (v >= 4294967295.0) return 4294967295u;
    return (uint32_t)v;
}

void lbo_band_table(double sample_rate, uint32_t window, uint32_t n_frames, uint32_t bands,
                    uint32_t* indices, uint32_t* lo, uint32_t* hi) {
    const double max_freq = sample_rate / 2.0;                       /* :362 */
    const double min_freq = 318.0;                                   /* :363 */
    const double log_base = exp(log(max_freq / min_freq) / bands);   /* :365 */
    const double mincoef = (double)window / sample_rate * min_freq;  /* :366 */
    for (uint32_t j = 0; j <= bands; ++j) {                          /* :368-371 */
        const uint32_t start = trunc_u32((pow(log_base, (double)j) - 1.0) * mincoef);
        indices[j] = start + trunc_u32(mincoef);
    }
    const double hz_per_bin = sample_rate / n_frames;                /* :382-383 */
    for (uint32_t i = 0; i < bands; ++i) {
        const uint32_t lb = indices[i], hb = indices[i + 1];
        uint32_t l = trunc_u32(((double)(uint32_t)(2u * lb)) / hz_per_bin - 1.0);
        uint32_t h = trunc_u32(((double)(uint32_t)(2u * hb)) / hz_per_bin - 1.0);
        /* the reference would read past the W-sample buffer for bins >= n_frames/2 */
        if (l > n_frames / 2) l = n_frames / 2;
        if (h > n_frames / 2) h = n_frames / 2;
        lo[i] = l;
        hi[i] = h;
    }
}

void lbo_band_energies(const float* spectrum, uint32_t n_frames, uint32_t bands,
                       const uint32_t* indices, const uint32_t* lo, const uint32_t* hi,
                       float* out) {
    const uint32_t width = (uint32_t)(n_frames / 2.0);               /* :373 */
    const float norm = (float)(width / 2);                           /* :391,394 */
    for (uint32_t i = 0; i < bands; ++i) {
        float p = 0.0f;
        for (uint32_t k = lo[i]; k < hi[i]; ++k) {
            float re = spectrum[2 * k];
            float im = spectrum[2 * k + 1];
            if (re > 0.0f) re /= norm;                               /* only positives: :390-395 */
            if (im > 0.0f) im /= norm;
            const float rr = re * re, ii = im * im;
            const float v = rr + ii;
            if (v == v && isfinite(v)) p += v;                       /* :398-401 */
        }
        out[i] = p / (float)(indices[i + 1] - indices[i]);           /* :404 */
    }
}

typedef struct row_ctx {
    fft_plan plan;
    uint32_t bands;
    uint32_t* indices;
    uint32_t* lo;
    uint32_t* hi;
    float* zr;
    float* spec;
} row_ctx;

static int row_ctx_init(row_ctx* c, const lbo_config* cfg) {
    if (cfg->bands == 0 || cfg->stride == 0) return -1;
    if (plan_init(&c->plan, cfg->window)) return -1;
    c->bands = cfg->bands;
    c->indices = (uint32_t*)malloc(sizeof(uint32_t) * (3 * cfg->bands + 1));
    c->lo = c->indices + cfg->bands + 1;
    c->hi = c->lo + cfg->bands;
    lbo_band_table(cfg->sample_rate, cfg->window, cfg->window, cfg->bands, c->indices, c->lo, c->hi);
    c->zr = (float*)malloc(sizeof(float) * cfg->window);
    c->spec = (float*)malloc(sizeof(float) * cfg->window);
    return 0;
}

static void row_ctx_free(row_ctx* c) {
    plan_free(&c->plan);
    free(c->indices);
    free(c->zr);
    free(c->spec);
}

static void row_ctx_run(row_ctx* c, const float* pcm, float* out_row) {
    rfft_exec(&c->plan, pcm, c->zr, c->zr + c->plan.N, c->spec);
    lbo_band_energies(c->spec, c->plan.W, c->bands, c->indices, c->lo, c->hi, out_row);
}

int lbo_window_row(const float* window_pcm, const lbo_config* cfg, float* out_row) {
    row_ctx c;
    if (row_ctx_init(&c, cfg)) return -1;
    row_ctx_run(&c, window_pcm, out_row);
    row_ctx_free(&c);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Haar -- LBAudioDetectiveFrame.m:113-153
 * ---------------------------------------------------------------------------------------- */
void lbo_haar_1d(float* a, uint32_t n) {
    if (n == 0) return;
    const float root = sqrtf((float)n);                              /* :137-139 */
    for (uint32_t i = 0; i < n; ++i) a[i] /= root;
    const float root2 = sqrtf(2.0f);
    float* tmp = (float*)malloc(sizeof(float) * n);
    uint32_t cnt = n;
    while (cnt > 1) {                                                /* :143-152 */
        cnt /= 2;
        for (uint32_t i = 0; i < cnt; ++i) {
            const float e = a[2 * i], o = a[2 * i + 1];
            tmp[i] = (e + o) / root2;
            tmp[cnt + i] = (e - o) / root2;
        }
        memcpy(a, tmp, sizeof(float) * 2 * cnt);
    }
    free(tmp);
}

void lbo_haar_2d(float* m, uint32_t rows, uint32_t cols) {
    for (uint32_t r = 0; r < rows; ++r) lbo_haar_1d(m + (size_t)r * cols, cols);   /* :114-116 */
    float* col = (float*)malloc(sizeof(float) * (rows ? rows : 1));
    for (uint32_t c = 0; c < cols; ++c) {                                          /* :118-131 */
        for (uint32_t r = 0; r < rows; ++r) col[r] = m[(size_t)r * cols + c];
        lbo_haar_1d(col, rows);
        for (uint32_t r = 0; r < rows; ++r) m[(size_t)r * cols + c] = col[r];
    }
    free(col);
}

/* ------------------------------------------------------------------------------------------
 * sign extraction -- LBAudioDetectiveFrame.m:165-191
 * ---------------------------------------------------------------------------------------- */
static int cmp_u64_desc(const void* pa, const void* pb) {
    const uint64_t a = *(const uint64_t*)pa, b = *(const uint64_t*)pb;
    return (a < b) - (a > b);
}

void lbo_extract(const float* m, uint32_t rows, uint32_t cols, uint32_t n_wavelets, uint8_t* out) {
    const uint32_t total = rows * cols;
    memset(out, 0, (size_t)2 * n_wavelets);
    if (total == 0) return;
    uint64_t* keys = (uint64_t*)malloc(sizeof(uint64_t) * total);
    for (uint32_t i = 0; i < total; ++i) {
        uint32_t bits;
        memcpy(&bits, &m[i], 4);
        /* |v| as an order-preserving integer; ~i makes the lower index win among equals */
        keys[i] = ((uint64_t)(bits & 0x7fffffffu) << 32) | (uint32_t)(~i);
    }
    qsort(keys, total, sizeof(uint64_t), cmp_u64_desc);              /* :176-178 */
    const uint32_t take = n_wavelets < total ? n_wavelets : total;
    for (uint32_t i = 0; i < take; ++i) {                            /* :182-190 */
        const uint32_t idx = ~(uint32_t)keys[i];
        const float v = m[idx];
        if (v > 0.0f) out[2 * i] = 1;
        else if (v < 0.0f) out[2 * i + 1] = 1;
    }
    free(keys);
}

/* ------------------------------------------------------------------------------------------
 * framing + synthesis -- LBAudioDetective.m:241-293,315-331
 * ---------------------------------------------------------------------------------------- */
uint64_t lbo_subfingerprint_count(uint64_t n_samples, uint32_t window, uint32_t stride) {
    if (stride == 0 || n_samples < window) return 0;
    const uint64_t image_width = (n_samples - window) / stride;     /* :250 */
    return image_width / LBO_ROWS_PER_FRAME;                         /* :255 */
}

static int config_ok(const lbo_config* cfg) {
    if (!is_pow2(cfg->window) || cfg->window < 8) return 0;
    if (cfg->stride == 0 || cfg->bands == 0) return 0;
    if (!(cfg->sample_rate > 0.0)) return 0;
    /* Extract is asked for subfp_len wavelets (LBAudioDetective.m:324) out of 128*bands */
    if ((uint64_t)cfg->subfp_len > (uint64_t)LBO_ROWS_PER_FRAME * cfg->bands) return 0;
    return 1;
}

uint64_t lbo_fingerprint_pcm_taps(const float* pcm, uint64_t n_samples, const lbo_config* cfg,
                                  uint8_t* out_bools, float* frames_raw, float* frames_haar) {
    if (!config_ok(cfg)) return (uint64_t)-1;
    const uint64_t count = lbo_subfingerprint_count(n_samples, cfg->window, cfg->stride);
    if (count == 0) return 0;
    row_ctx rc;
    if (row_ctx_init(&rc, cfg)) return (uint64_t)-1;
    const size_t frame_elems = (size_t)LBO_ROWS_PER_FRAME * cfg->bands;
    float* frame = (float*)malloc(sizeof(float) * frame_elems);
    uint8_t* pairs = (uint8_t*)malloc((size_t)2 * cfg->subfp_len);
    for (uint64_t f = 0; f < count; ++f) {
        for (uint32_t r = 0; r < LBO_ROWS_PER_FRAME; ++r) {          /* :262-290 */
            const uint64_t win = f * LBO_ROWS_PER_FRAME + r;
            row_ctx_run(&rc, pcm + win * cfg->stride, frame + (size_t)r * cfg->bands);
        }
        if (frames_raw) memcpy(frames_raw + f * frame_elems, frame, sizeof(float) * frame_elems);
        lbo_haar_2d(frame, LBO_ROWS_PER_FRAME, cfg->bands);          /* :320 */
        if (frames_haar) memcpy(frames_haar + f * frame_elems, frame, sizeof(float) * frame_elems);
        lbo_extract(frame, LBO_ROWS_PER_FRAME, cfg->bands, cfg->subfp_len, pairs);   /* :324 */
        /* AddSubfingerprint keeps the first subfp_len Booleans only (Fingerprint.m:91-94) */
        memcpy(out_bools + f * cfg->subfp_len, pairs, cfg->subfp_len);
    }
    free(pairs);
    free(frame);
    row_ctx_free(&rc);
    return count;
}

/* ------------------------------------------------------------------------------------------
 * the file loop as upstream really runs it -- LBAudioDetective.m:236-293
 *
 * dataLength (:236) and the seek offsets (:287-288) are FILE frames, each read (:275) asks for
 * CLIENT frames (processing rate), so with a 44.1 kHz file and the 5512 Hz client format
 *   - the window count comes from the file length: imageWidth = (file_frames - W) / stride (:250);
 *   - window i starts `hop` client samples after window i-1 (hop = stride * rate / file_rate);
 *   - readNumberFrames is an in/out argument declared outside the loop (:252,275): a short read near
 *     the end of the file shrinks every later request as well;
 *   - the FFT runs in place in `samples` (:351-355), so the W - nRead floats a short read leaves
 *     untouched are the PREVIOUS window's packed spectrum, and the FFT still spans all W floats;
 *   - ComputeFrequencies is handed nRead as inNumberFrames (:281): width, the positive-only
 *     normalisation and the band bin bounds (:373,382-383,390-395) all follow nRead.
 * `client` is the whole file already converted to the processing rate.  Before the first read the
 * buffer is taken as zeros (upstream: uninitialised stack).
 * ---------------------------------------------------------------------------------------- */
uint64_t lbo_fingerprint_file_loop(const float* client, uint64_t n_client, uint64_t file_frames,
                                   uint32_t hop, int tail_mode, const lbo_config* cfg,
                                   uint8_t* out_bools, float* frames_raw, uint32_t* out_n_read) {
    if (!config_ok(cfg) || hop == 0 || tail_mode < 0 || tail_mode > 2) return (uint64_t)-1;
    if (file_frames < cfg->window) return 0;
    const uint64_t image_width = (file_frames - cfg->window) / cfg->stride;      /* :250 */
    const uint64_t count = image_width / LBO_ROWS_PER_FRAME;                      /* :255 */
    if (count == 0) return 0;
    row_ctx rc;
    if (row_ctx_init(&rc, cfg)) return (uint64_t)-1;
    const uint32_t W = cfg->window;
    const size_t frame_elems = (size_t)LBO_ROWS_PER_FRAME * cfg->bands;
    float* frame = (float*)malloc(sizeof(float) * frame_elems);
    uint8_t* pairs = (uint8_t*)malloc((size_t)2 * cfg->subfp_len);
    float* samples = (float*)calloc(W, sizeof(float));                           /* :243 */
    uint32_t* lo = (uint32_t*)malloc(sizeof(uint32_t) * 2 * cfg->bands);
    uint32_t* hi = lo + cfg->bands;
    uint32_t n_read = W;                                                         /* :252 */
    for (uint64_t f = 0; f < count; ++f) {
        for (uint32_t r = 0; r < LBO_ROWS_PER_FRAME; ++r) {                      /* :262-290 */
            const uint64_t i = f * LBO_ROWS_PER_FRAME + r;
            const uint64_t start = i * (uint64_t)hop;                            /* :287-288 */
            const uint64_t avail = n_client > start ? n_client - start : 0;
            float* row = frame + (size_t)r * cfg->bands;
            if (tail_mode == LBO_TAIL_NOTHING && avail < W) {
                /* a read that cannot be met in full delivers 0 frames (and, readNumberFrames being
                 * in/out, so does every later one): inNumberFrames == 0 makes both bin bounds of
                 * every band (UInt32)(0 - 1.0) == 0 on ARM (:382-383), the loops are empty, the row
                 * is 0 / width (:404).  The FFT of the stale buffer still runs but is never read. */
                n_read = 0;
                if (out_n_read) out_n_read[i] = 0;
                for (uint32_t b = 0; b < cfg->bands; ++b) row[b] = 0.0f / (float)(rc.indices[b + 1] - rc.indices[b]);
                continue;
            }
            if (tail_mode == LBO_TAIL_ZERO_FILL) {
                /* not upstream: the unread part of the window is cleared */
                memset(samples, 0, sizeof(float) * W);
                n_read = W;
                memcpy(samples, client + (avail ? start : 0), sizeof(float) * (avail < W ? avail : W));
                if (out_n_read) out_n_read[i] = W;
                rfft_exec(&rc.plan, samples, rc.zr, rc.zr + rc.plan.N, rc.spec);
                lbo_band_energies(rc.spec, W, cfg->bands, rc.indices, rc.lo, rc.hi, row);
                continue;
            }
            if (avail < n_read) n_read = (uint32_t)avail;                        /* :275, in/out */
            memcpy(samples, client + (avail ? start : 0), sizeof(float) * n_read);
            if (out_n_read) out_n_read[i] = n_read;
            rfft_exec(&rc.plan, samples, rc.zr, rc.zr + rc.plan.N, rc.spec);     /* :353-355 */
            memcpy(samples, rc.spec, sizeof(float) * W);                         /* in place */
            const float* spec = samples;
            if (n_read == W) {
                lbo_band_energies(spec, W, cfg->bands, rc.indices, rc.lo, rc.hi, row);
            } else {
                /* bin bounds from nRead (:382-383); reads stay inside the W-float buffer */
                const double hz_per_bin = cfg->sample_rate / (double)n_read;
                for (uint32_t b = 0; b < cfg->bands; ++b) {
                    uint32_t l = trunc_u32(((double)(uint32_t)(2u * rc.indices[b])) / hz_per_bin - 1.0);
                    uint32_t h = trunc_u32(((double)(uint32_t)(2u * rc.indices[b + 1])) / hz_per_bin - 1.0);
                    if (l > W / 2) l = W / 2;
                    if (h > W / 2) h = W / 2;
                    lo[b] = l;
                    hi[b] = h;
                }
                lbo_band_energies(spec, n_read, cfg->bands, rc.indices, lo, hi, row);
            }
        }
        if (frames_raw) memcpy(frames_raw + f * frame_elems, frame, sizeof(float) * frame_elems);
        lbo_haar_2d(frame, LBO_ROWS_PER_FRAME, cfg->bands);
        lbo_extract(frame, LBO_ROWS_PER_FRAME, cfg->bands, cfg->subfp_len, pairs);
        memcpy(out_bools + f * cfg->subfp_len, pairs, cfg->subfp_len);
    }
    free(lo);
    free(samples);
    free(pairs);
    free(frame);
    row_ctx_free(&rc);
    return count;
}

/* Stage splits used by tools/vdsp_gap_probe.py: the pipeline behind an FFT that is NOT the canonical one
 * (spectra in the packed layout of lbo_rfft_packed), and the frame stage alone. */
int lbo_spectra_to_rows(const float* spectra, uint64_t n_windows, const lbo_config* cfg, float* rows) {
    if (!config_ok(cfg)) return -1;
    row_ctx rc;
    if (row_ctx_init(&rc, cfg)) return -1;
    for (uint64_t w = 0; w < n_windows; ++w)
        lbo_band_energies(spectra + w * cfg->window, cfg->window, cfg->bands, rc.indices, rc.lo, rc.hi,
                          rows + w * cfg->bands);
    row_ctx_free(&rc);
    return 0;
}

int lbo_rows_to_subfingerprints(const float* rows, uint64_t n_frames, const lbo_config* cfg, uint8_t* out_bools) {
    if (!config_ok(cfg)) return -1;
    const size_t frame_elems = (size_t)LBO_ROWS_PER_FRAME * cfg->bands;
    float* frame = (float*)malloc(sizeof(float) * frame_elems);
    uint8_t* pairs = (uint8_t*)malloc((size_t)2 * cfg->subfp_len);
    for (uint64_t f = 0; f < n_frames; ++f) {
        memcpy(frame, rows + f * frame_elems, sizeof(float) * frame_elems);
        lbo_haar_2d(frame, LBO_ROWS_PER_FRAME, cfg->bands);
        lbo_extract(frame, LBO_ROWS_PER_FRAME, cfg->bands, cfg->subfp_len, pairs);
        memcpy(out_bools + f * cfg->subfp_len, pairs, cfg->subfp_len);
    }
    free(pairs);
    free(frame);
    return 0;
}

int lbo_rfft_packed_batch(const float* x, uint64_t n_windows, uint32_t W, float* out) {
    fft_plan p;
    if (plan_init(&p, W)) return -1;
    float* z = (float*)malloc(sizeof(float) * W);
    for (uint64_t w = 0; w < n_windows; ++w) rfft_exec(&p, x + w * W, z, z + p.N, out + w * W);
    free(z);
    plan_free(&p);
    return 0;
}

uint64_t lbo_fingerprint_pcm(const float* pcm, uint64_t n_samples, const lbo_config* cfg,
                             uint8_t* out_bools) {
    return lbo_fingerprint_pcm_taps(pcm, n_samples, cfg, out_bools, NULL, NULL);
}

int lbo_fingerprint_batch(const float* pcm, uint64_t n_clips, uint64_t samples_per_clip,
                          const lbo_config* cfg, uint8_t* out_bools, int nthreads) {
    if (!config_ok(cfg)) return -1;
    const uint64_t per = lbo_subfingerprint_count(samples_per_clip, cfg->window, cfg->stride);
    int failed = 0;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int64_t c = 0; c < (int64_t)n_clips; ++c) {
        const uint64_t got = lbo_fingerprint_pcm(pcm + (uint64_t)c * samples_per_clip, samples_per_clip,
                                                 cfg, out_bools + (uint64_t)c * per * cfg->subfp_len);
        if (got != per) failed = 1;
    }
    return failed ? -1 : 0;
}

/* ------------------------------------------------------------------------------------------
 * compare -- LBAudioDetectiveFingerprint.m:119-176
 * ---------------------------------------------------------------------------------------- */
float lbo_compare_sub(const uint8_t* a, const uint8_t* b, uint32_t subfp_len, uint32_t range) {
    uint32_t possible = 0, hits = 0;
    const uint32_t lim = range < subfp_len ? range : subfp_len;      /* :155 */
    for (uint32_t i = 0; i < lim; i += 2) {
        /* an odd limit makes the reference read element lim of a lim-sized array; the
         * oracle treats the missing Boolean as 0 */
        const uint8_t a0 = a[i], a1 = (i + 1 < subfp_len) ? a[i + 1] : 0;
        if (a0 || a1) {                                              /* :159 */
            ++possible;
            const uint8_t b0 = b[i], b1 = (i + 1 < subfp_len) ? b[i + 1] : 0;
            if (a0 == b0 && a1 == b1) ++hits;                        /* :165 */
        }
    }
    if (possible == 0) return 0.0f;                                  /* :171-173 */
    return (float)hits / (float)possible;
}

float lbo_compare_fp(const uint8_t* fp1, uint32_t n1, const uint8_t* fp2, uint32_t n2,
                     uint32_t subfp_len, uint32_t range) {
    if (n1 < n2) {                                                   /* :123-131 */
        const uint8_t* t = fp1; fp1 = fp2; fp2 = t;
        const uint32_t u = n1; n1 = n2; n2 = u;
    }
    float match = 0.0f;
    for (uint32_t offset = 0; offset <= n1 - n2; ++offset) {         /* :136 */
        float sum = 0.0f;
        for (uint32_t i = 0; i < n2; ++i)
            sum += lbo_compare_sub(fp1 + (size_t)(i + offset) * subfp_len,
                                   fp2 + (size_t)i * subfp_len, subfp_len, range);
        const float cand = sum / (float)n2;
        /* Foundation's MAX(A,B) is (a < b ? b : a): a NaN candidate (n2 == 0) is dropped */
        match = (match < cand) ? cand : match;                       /* :144 */
    }
    return match;
}

void lbo_corpus_best(const uint8_t* query, uint32_t n_query, const uint8_t* corpus,
                     uint64_t n_entries, uint32_t n_sub, uint32_t subfp_len, uint32_t range,
                     int nthreads, int64_t* best_index, float* best_score) {
    float* scores = (float*)malloc(sizeof(float) * (n_entries ? n_entries : 1));
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int64_t e = 0; e < (int64_t)n_entries; ++e)
        scores[e] = lbo_compare_fp(query, n_query, corpus + (size_t)e * n_sub * subfp_len, n_sub,
                                   subfp_len, range);
    float best = 0.0f;                                               /* Tests.m:60 */
    int64_t idx = -1;
    for (uint64_t e = 0; e < n_entries; ++e)
        if (best < scores[e]) { best = scores[e]; idx = (int64_t)e; }   /* Tests.m:80-83 */
    free(scores);
    *best_index = idx;
    *best_score = best;
}

/* The same loop over a corpus whose entries have their own sub-fingerprint counts -- the shape upstream's test
 * really has (LBAudioDetectiveTests.m:57-91: ten sequences of different lengths; Fingerprint.m:123-146 swaps and
 * slides).  corpus = the entries' Booleans back to back, counts[e] sub-fingerprints each; scores_out (optional)
 * receives every entry's match. */
void lbo_corpus_best_ragged(const uint8_t* query, uint32_t n_query, const uint8_t* corpus, const uint32_t* counts,
                            uint64_t n_entries, uint32_t subfp_len, uint32_t range, int nthreads,
                            int64_t* best_index, float* best_score, float* scores_out) {
    float* scores = scores_out ? scores_out : (float*)malloc(sizeof(float) * (n_entries ? n_entries : 1));
    uint64_t* start = (uint64_t*)malloc(sizeof(uint64_t) * (n_entries + 1));
    start[0] = 0;
    for (uint64_t e = 0; e < n_entries; ++e) start[e + 1] = start[e] + counts[e];
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int64_t e = 0; e < (int64_t)n_entries; ++e)
        scores[e] = lbo_compare_fp(query, n_query, corpus + (size_t)start[e] * subfp_len, counts[e], subfp_len, range);
    float best = 0.0f;                                               /* Tests.m:60 */
    int64_t idx = -1;
    for (uint64_t e = 0; e < n_entries; ++e)
        if (best < scores[e]) { best = scores[e]; idx = (int64_t)e; }   /* Tests.m:80-83 */
    free(start);
    if (!scores_out) free(scores);
    *best_index = idx;
    *best_score = best;
}

/* ------------------------------------------------------------------------------------------
 * The same compare on PACKED sub-fingerprints: the second CPU baseline SURVEY 8(d) asks for ("the packed popcount
 * CPU version").  Boolean b of a sub-fingerprint is bit b & 63 of word b >> 6 (four 64-bit words, subfp_len <= 256).
 * With A the first argument's sub-fingerprint (Fingerprint.m:151-176 on bit pairs, the identity SURVEY a-11 verified
 * against the compiled reference):  NZ = (A | A >> 1) & EVEN & RANGE;  possible = popc(NZ);
 * Y = (A ^ B) | (A ^ B) >> 1;  hits = popc(NZ & ~Y).  A pair never straddles a word (pairs start at even bits).
 * Checked against lbo_corpus_best by tests/test_oracle.py.
 * ---------------------------------------------------------------------------------------- */
void lbo_pack_bools(const uint8_t* bools, uint64_t n_rows, uint32_t subfp_len, uint64_t* out) {
    for (uint64_t r = 0; r < n_rows; ++r) {
        uint64_t w[4] = {0, 0, 0, 0};
        const uint8_t* row = bools + (size_t)r * subfp_len;
        for (uint32_t b = 0; b < subfp_len && b < 256u; ++b)
            if (row[b]) w[b >> 6] |= 1ull << (b & 63u);
        memcpy(out + 4 * r, w, sizeof w);
    }
}

static inline float packed_compare_sub(const uint64_t* a, const uint64_t* b, const uint64_t* range_mask) {
    uint32_t possible = 0, hits = 0;
    for (int w = 0; w < 4; ++w) {
        const uint64_t nz = (a[w] | (a[w] >> 1)) & range_mask[w];
        const uint64_t x = a[w] ^ b[w];
        possible += (uint32_t)__builtin_popcountll(nz);
        hits += (uint32_t)__builtin_popcountll(nz & ~(x | (x >> 1)));
    }
    if (possible == 0) return 0.0f;                                  /* Fingerprint.m:171-173 */
    return (float)hits / (float)possible;
}

static float packed_compare_fp(const uint64_t* fp1, uint32_t n1, const uint64_t* fp2, uint32_t n2,
                               const uint64_t* range_mask) {
    if (n1 < n2) {                                                   /* Fingerprint.m:123-131 */
        const uint64_t* t = fp1; fp1 = fp2; fp2 = t;
        const uint32_t u = n1; n1 = n2; n2 = u;
    }
    float match = 0.0f;
    for (uint32_t offset = 0; offset <= n1 - n2; ++offset) {         /* :136 */
        float sum = 0.0f;
        for (uint32_t i = 0; i < n2; ++i)
            sum += packed_compare_sub(fp1 + 4 * (size_t)(i + offset), fp2 + 4 * (size_t)i, range_mask);
        const float cand = sum / (float)n2;
        match = (match < cand) ? cand : match;                       /* :144 */
    }
    return match;
}

void lbo_corpus_best_packed(const uint64_t* query, uint32_t n_query, const uint64_t* corpus,
                            uint64_t n_entries, uint32_t n_sub, uint32_t subfp_len, uint32_t range,
                            int nthreads, int64_t* best_index, float* best_score) {
    /* EVEN & RANGE: the even bits below min(range, subfp_len) (Fingerprint.m:155; pairs i = 0, 2, ...) */
    uint64_t mask[4] = {0, 0, 0, 0};
    const uint32_t lim = range < subfp_len ? range : subfp_len;
    for (uint32_t i = 0; i < lim && i < 256u; i += 2) mask[i >> 6] |= 1ull << (i & 63u);
    float* scores = (float*)malloc(sizeof(float) * (n_entries ? n_entries : 1));
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int64_t e = 0; e < (int64_t)n_entries; ++e)
        scores[e] = packed_compare_fp(query, n_query, corpus + 4 * (size_t)e * n_sub, n_sub, mask);
    float best = 0.0f;                                               /* Tests.m:60 */
    int64_t idx = -1;
    for (uint64_t e = 0; e < n_entries; ++e)
        if (best < scores[e]) { best = scores[e]; idx = (int64_t)e; }   /* Tests.m:80-83 */
    free(scores);
    *best_index = idx;
    *best_score = best;
}

/* ------------------------------------------------------------------------------------------
 * synthetic inputs (integer arithmetic only, so a device generator can match bit for bit)
 * ---------------------------------------------------------------------------------------- */
static inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

void lbo_synth_sine_table(int16_t* t) {
    /* quarter-wave symmetric so the table does not depend on libm's last bit */
    for (int i = 0; i <= 256; ++i) {
        const double s = sin((2.0 * M_PI * i) / 1024.0) * 32767.0;
        const int16_t v = (int16_t)floor(s + 0.5);
        t[i] = v;
        if (i > 0 && i < 256) {
            t[512 - i] = v;
            t[512 + i] = (int16_t)-v;
            t[1024 - i] = (int16_t)-v;
        }
    }
    t[512] = 0;
    t[768] = (int16_t)-t[256];
}

static int32_t synth_channel(const int16_t* sine, uint32_t key, uint32_t rate_hz, uint32_t n) {
    int32_t acc = (int32_t)(mix32(key ^ (n * 0x9E3779B1u)) >> 18) - 8192;   /* +-0.25 full scale */
    for (uint32_t j = 0; j < 3; ++j) {
        const uint32_t f_mhz = 60000u + mix32(key + 11u * j + 1u) % 840001u;   /* 60..900 Hz */
        const uint32_t step = (uint32_t)((((uint64_t)f_mhz) << 32) / ((uint64_t)rate_hz * 1000u));
        const uint32_t phase0 = mix32(key + 11u * j + 2u);
        const int32_t amp = 1638 + (int32_t)(mix32(key + 11u * j + 3u) % 8193u);   /* 0.05..0.3 */
        const uint32_t ph = phase0 + n * step;
        acc += (amp * (int32_t)sine[ph >> 22]) >> 15;
    }
    if (acc > 32767) acc = 32767;
    if (acc < -32768) acc = -32768;
    return acc;
}

void lbo_synth_clip(uint32_t seed, uint64_t clip, double sample_rate, uint32_t n_samples,
                    int stereo_sum, float* out) {
    int16_t sine[1024];
    lbo_synth_sine_table(sine);
    const uint32_t rate_hz = (uint32_t)sample_rate;
    const uint32_t key = mix32(seed ^ mix32((uint32_t)clip) ^ (uint32_t)(clip >> 32) * 0x632BE5ABu);
    for (uint32_t n = 0; n < n_samples; ++n) {
        if (stereo_sum) {
            const int32_t l = synth_channel(sine, key, rate_hz, n);
            const int32_t r = synth_channel(sine, key ^ 0x5bd1e995u, rate_hz, n);
            out[n] = (float)(l + r) / 65536.0f;   /* 0.5 * (L + R), exact */
        } else {
            out[n] = (float)synth_channel(sine, key, rate_hz, n) / 32768.0f;
        }
    }
}

void lbo_synth_entry(uint32_t seed, uint64_t entry, uint32_t n_sub, uint32_t subfp_len,
                     uint8_t* out) {
    const uint32_t key = mix32(seed ^ mix32((uint32_t)entry) ^ (uint32_t)(entry >> 32) * 0x632BE5ABu);
    const uint32_t pairs = (subfp_len + 1) / 2;
    for (uint32_t s = 0; s < n_sub; ++s) {
        uint8_t* row = out + (size_t)s * subfp_len;
        for (uint32_t p = 0; p < pairs; ++p) {
            const uint32_t r = mix32(key + (s * 1024u + p) * 0x9E3779B1u);
            uint8_t pos = 0, neg = 0;
            if (r % 100u != 0u) {          /* 1 % of pairs are "00" */
                if ((r >> 8) & 1u) pos = 1; else neg = 1;
            }
            row[2 * p] = pos;
            if (2 * p + 1 < subfp_len) row[2 * p + 1] = neg;
        }
    }
}

/* sub-fingerprint count of entry `entry` of the synthetic ragged corpus: lo..hi, uniform */
uint32_t lbo_synth_ragged_count(uint32_t seed, uint64_t entry, uint32_t lo, uint32_t hi) {
    const uint32_t r = mix32(seed ^ 0x52414747u ^ mix32((uint32_t)entry) ^ (uint32_t)(entry >> 32) * 0x632BE5ABu);
    return lo + r % (hi - lo + 1u);
}
