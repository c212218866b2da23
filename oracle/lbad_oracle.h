/*
 * lbad_oracle.h -- CPU parity oracle for the LBAudioDetective hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
 * there only as the checker.  The shipped path is the HIP library built from
 * lbaudiodetective_amd/csrc (it never links or calls this file).
 *
 * What it restates (paths relative to the upstream reference tree):
 *   framing            LBAudioDetective/LBAudioDetective.m:241-293
 *   FFT packing+bands  LBAudioDetective/LBAudioDetective.m:351-405
 *   frame / Haar       LBAudioDetective/LBAudioDetectiveFrame.m:86-153
 *   sign extraction    LBAudioDetective/LBAudioDetectiveFrame.m:165-191
 *   synthesis          LBAudioDetective/LBAudioDetective.m:315-331
 *   compare            LBAudioDetective/LBAudioDetectiveFingerprint.m:119-176
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - Haar leg: pinned by the reference's own known answer (test matrix at
 *     LBAudioDetectiveTests/LBAudioDetectiveTests.m:160-162, expected output in the
 *     essay's Fig. 13) -- tests/test_oracle.py::test_haar_known_answer.
 *   - Band-edge tables: pinned against the tables the survey derived from the
 *     reference arithmetic.
 *   - FFT leg: the reference calls Apple vDSP (closed source, absent here).  The
 *     oracle restates the documented contract of vDSP_fft_zrip (radix-2, forward,
 *     output = 2 x DFT, DC/Nyquist packed in element 0) with one fixed float32
 *     operation order.  PARITY UNPINNED at the vDSP boundary: no golden vector of
 *     vDSP output exists in the reference.
 *   - Sort ties: NSMutableArray sort order for equal keys is unspecified by the
 *     API; the oracle uses ascending original index.  PARITY UNPINNED for ties.
 *   - Compare leg: pure integer/float32 C in the reference, restated line for
 *     line; the reference holds no golden vectors for it and cannot be compiled
 *     here without stand-in Apple headers, so it is pinned only through the
 *     invariants the reference's tests use (copy == original, self-compare == 1).
 */
#ifndef LBAD_ORACLE_H
#define LBAD_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LBO_ROWS_PER_FRAME 128u /* LBAudioDetective.m:25 */

typedef struct lbo_config {
    double sample_rate;   /* processingFormat.mSampleRate, default 5512  (LBAudioDetective.m:128) */
    uint32_t window;      /* windowSize, default 2048                    (LBAudioDetective.m:22)  */
    uint32_t stride;      /* analysisStride, default 64                  (LBAudioDetective.m:23)  */
    uint32_t bands;       /* pitchStepCount, default 32                  (LBAudioDetective.m:24)  */
    uint32_t subfp_len;   /* subfingerprintLength, default 200           (LBAudioDetective.m:26)  */
} lbo_config;

void lbo_default_config(lbo_config* cfg);

/* Master twiddle table: tw[k] = exp(-2*pi*i*k/W) for k in [0, W/2), float32,
 * generated in double precision with exact octant symmetry. */
void lbo_twiddles(uint32_t W, float* re, float* im);

/* Canonical forward real FFT of W float32 samples (W a power of two >= 8).
 * Output layout = vDSP_fft_zrip + vDSP_ztoc (LBAudioDetective.m:353-355):
 * out[0] = 2*DC, out[1] = 2*Nyquist, out[2k], out[2k+1] = 2*Re X[k], 2*Im X[k]. */
int lbo_rfft_packed(const float* x, uint32_t W, float* out);

/* Band-edge arithmetic of LBAudioDetective.m:361-371,382-383.
 * indices has bands+1 entries; lo/hi have bands entries (bin index bounds, hi clamped
 * to n_frames/2). */
void lbo_band_table(double sample_rate, uint32_t window, uint32_t n_frames, uint32_t bands,
                    uint32_t* indices, uint32_t* lo, uint32_t* hi);

/* Band means of LBAudioDetective.m:373-405 over a packed spectrum. */
void lbo_band_energies(const float* spectrum, uint32_t n_frames, uint32_t bands,
                       const uint32_t* indices, const uint32_t* lo, const uint32_t* hi,
                       float* out);

/* One analysis window: FFT + bands (LBAudioDetectiveComputeFrequencies). */
int lbo_window_row(const float* window_pcm, const lbo_config* cfg, float* out_row);

/* LBAudioDetectiveFrameDecomposeArray (Frame.m:134-153) on a contiguous array. */
void lbo_haar_1d(float* a, uint32_t n);
/* LBAudioDetectiveFrameDecompose (Frame.m:113-132): every row, then every column;
 * m is row-major rows x cols. */
void lbo_haar_2d(float* m, uint32_t rows, uint32_t cols);

/* LBAudioDetectiveFrameExtractFingerprint (Frame.m:165-191): rank all rows*cols
 * coefficients by |v| descending (ties: ascending flat index) and write the sign pair
 * of rank i to out[2i], out[2i+1].  out has 2*n_wavelets entries, zeroed first. */
void lbo_extract(const float* m, uint32_t rows, uint32_t cols, uint32_t n_wavelets, uint8_t* out);

/* Number of sub-fingerprints a PCM buffer of n_samples yields (LBAudioDetective.m:250-255;
 * n_samples < window yields 0 instead of the reference's unsigned wrap). */
uint64_t lbo_subfingerprint_count(uint64_t n_samples, uint32_t window, uint32_t stride);

/* Whole fingerprint leg on PCM already at the processing rate.  out_bools receives
 * count * cfg->subfp_len Booleans; returns count, or (uint64_t)-1 for an invalid config. */
uint64_t lbo_fingerprint_pcm(const float* pcm, uint64_t n_samples, const lbo_config* cfg,
                             uint8_t* out_bools);

/* Optional taps for stage-level tests: the 128 x bands frame before and after the Haar. */
uint64_t lbo_fingerprint_pcm_taps(const float* pcm, uint64_t n_samples, const lbo_config* cfg,
                                  uint8_t* out_bools, float* frames_raw, float* frames_haar);

/* The file loop with upstream's file-frame bookkeeping and short reads at the end of the file
 * (LBAudioDetective.m:236-293): `client` is the file at the processing rate, file_frames
 * its length in FILE frames, hop the client samples between window starts.  frames_raw (optional)
 * receives the frames before the Haar, out_n_read (optional) nRead of every window used. */
#define LBO_TAIL_ZERO_FILL 0  /* unread part of a short window cleared (not upstream)                    */
#define LBO_TAIL_NOTHING   1  /* a read that cannot be met in full delivers 0 frames: all-zero rows       */
#define LBO_TAIL_STALE     2  /* partial reads; the unread part keeps the previous window's spectrum       */
uint64_t lbo_fingerprint_file_loop(const float* client, uint64_t n_client, uint64_t file_frames,
                                   uint32_t hop, int tail_mode, const lbo_config* cfg,
                                   uint8_t* out_bools, float* frames_raw, uint32_t* out_n_read);

/* Stage splits (tools/vdsp_gap_probe.py): packed spectra of n_windows windows -> band rows; 128-row frames ->
 * sub-fingerprints; the canonical FFT over a batch of windows. */
int lbo_spectra_to_rows(const float* spectra, uint64_t n_windows, const lbo_config* cfg, float* rows);
int lbo_rows_to_subfingerprints(const float* rows, uint64_t n_frames, const lbo_config* cfg, uint8_t* out_bools);
int lbo_rfft_packed_batch(const float* x, uint64_t n_windows, uint32_t W, float* out);

/* n_clips equal-length clips, nthreads OpenMP threads (cpu_baseline leg). */
int lbo_fingerprint_batch(const float* pcm, uint64_t n_clips, uint64_t samples_per_clip,
                          const lbo_config* cfg, uint8_t* out_bools, int nthreads);

/* LBAudioDetectiveFingerprintCompareSubfingerprints (Fingerprint.m:151-176). */
float lbo_compare_sub(const uint8_t* a, const uint8_t* b, uint32_t subfp_len, uint32_t range);
/* LBAudioDetectiveFingerprintCompareToFingerprint (Fingerprint.m:119-149).
 * fp1/fp2 are n x subfp_len Booleans, row-major. */
float lbo_compare_fp(const uint8_t* fp1, uint32_t n1, const uint8_t* fp2, uint32_t n2,
                     uint32_t subfp_len, uint32_t range);

/* Best-match loop of LBAudioDetectiveTests.m:57-91 scaled to a corpus: query is the fixed
 * first argument, every corpus entry the second.  Strict '<' from 0.0: lowest index wins
 * ties, best_index = -1 when no entry scores above 0. */
void lbo_corpus_best(const uint8_t* query, uint32_t n_query, const uint8_t* corpus,
                     uint64_t n_entries, uint32_t n_sub, uint32_t subfp_len, uint32_t range,
                     int nthreads, int64_t* best_index, float* best_score);

/* lbo_corpus_best for entries with their own sub-fingerprint counts (corpus = all entries' Booleans back to back);
 * scores_out (optional, n_entries floats) receives every entry's match. */
void lbo_corpus_best_ragged(const uint8_t* query, uint32_t n_query, const uint8_t* corpus, const uint32_t* counts,
                            uint64_t n_entries, uint32_t subfp_len, uint32_t range, int nthreads,
                            int64_t* best_index, float* best_score, float* scores_out);

/* Packed form of the same compare (SURVEY 8d's "packed popcount CPU version"): lbo_pack_bools turns n_rows rows of
 * subfp_len (<= 256) Booleans into four 64-bit words each (Boolean b = bit b & 63 of word b >> 6);
 * lbo_corpus_best_packed is lbo_corpus_best on such rows (query n_query x 4 words, corpus n_entries x n_sub x 4). */
void lbo_pack_bools(const uint8_t* bools, uint64_t n_rows, uint32_t subfp_len, uint64_t* out);
void lbo_corpus_best_packed(const uint64_t* query, uint32_t n_query, const uint64_t* corpus,
                            uint64_t n_entries, uint32_t n_sub, uint32_t subfp_len, uint32_t range,
                            int nthreads, int64_t* best_index, float* best_score);

/* Deterministic integer synthetic PCM (bench/test input, not a reference function). */
void lbo_synth_sine_table(int16_t* table1024);
void lbo_synth_clip(uint32_t seed, uint64_t clip, double sample_rate, uint32_t n_samples,
                    int stereo_sum, float* out);
/* Synthetic corpus entry: n_sub x subfp_len Booleans. */
void lbo_synth_entry(uint32_t seed, uint64_t entry, uint32_t n_sub, uint32_t subfp_len,
                     uint8_t* out);
/* Sub-fingerprint count of entry `entry` of the synthetic RAGGED corpus (uniform in lo..hi); its Booleans are
 * lbo_synth_entry(seed, entry, count, ...). */
uint32_t lbo_synth_ragged_count(uint32_t seed, uint64_t entry, uint32_t lo, uint32_t hi);

/* ---- file front end (lbad_file_oracle.c: own restatement of the container formats, the IMA4 / LPCM decoders and
 * the three documented converter models; stands in for ExtAudioFile, LBAudioDetective.m:224-237) ------------- */
/* Whole file -> mono float32 at the file's rate (channels averaged, CAF packet-table priming / valid-frame
 * trimming applied).  *out_mono is malloc'ed (lbo_file_free).  0 ok, -43 not found, 1 unsupported, 2 no memory. */
int lbo_file_decode(const char* path, float** out_mono, uint64_t* out_frames, double* out_rate);
int lbo_file_decode_bytes(const uint8_t* file, uint64_t n_bytes, float** out_mono, uint64_t* out_frames, double* out_rate);
void lbo_file_free(float* p);
/* probe only (tools/ima4_gap_probe.py): 1 = carry the running IMA4 predictor across packets when the packet header
 * agrees with it; 0 (default, the model everything is tested against) = every packet restarts from its header */
void lbo_file_set_ima4_carry(int on);
/* Converter models 0 (Kaiser sinc, 24 zero crossings, beta 9, cut-off 0.92), 1 (4 zero crossings, beta 3, cut-off
 * 1.0), 2 (linear interpolation); out holds lbo_resample_count() samples. */
uint64_t lbo_resample_count(uint64_t n_in, double rate_in, double rate_out);
int lbo_resample(const float* in, uint64_t n_in, double rate_in, double rate_out, int model, float* out);
/* decode + convert + upstream's window loop (hop_mode 1: file-frame bookkeeping with tail_mode; 0: PCM framing) */
int lbo_fingerprint_file(const char* path, const lbo_config* cfg, int hop_mode, int tail_mode, int resampler,
                         uint8_t** out_bools, uint64_t* out_count);

/* the same for n files, one per OpenMP thread (the CPU baseline of a catalogue build) */
int lbo_fingerprint_files(const char* const* paths, uint64_t n, const lbo_config* cfg, int hop_mode, int tail_mode, int resampler,
                          int nthreads, uint8_t** out_bools, uint64_t* out_counts);

#ifdef __cplusplus
}
#endif
#endif
