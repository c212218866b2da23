// CPU model of the dataflow of k_rows_stream.hip (W = 4096), first version: D4 per half, stage 5 with the sign
// folded into the twiddle, stage 6, one row per lane with the pruned split.  Scalar loops, fmaf; checks the rows of
// 128 windows bit for bit against oracle/lbad_oracle.c.   gcc -O2 -ffp-contract=off -Ioracle tools/exp/model_stream4096.c -Loracle/_build -llbad_oracle -lm
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "lbad_oracle.h"

#define W 4096
#define N 2048
static float twr[N], twi[N];
typedef struct { float x, y; } cplx;
static cplx cadd(cplx a, cplx b) { cplx r = {a.x + b.x, a.y + b.y}; return r; }
static cplx csub(cplx a, cplx b) { cplx r = {a.x - b.x, a.y - b.y}; return r; }
// u + w v, general form (oracle bfly_general "u" output)
static cplx madd(cplx u, float wr, float wi, cplx v) {
    cplx r = {fmaf(wr, v.x, fmaf(-wi, v.y, u.x)), fmaf(wr, v.y, fmaf(wi, v.x, u.y))};
    return r;
}
static cplx msub(cplx u, float wr, float wi, cplx v) {
    cplx r = {fmaf(-wr, v.x, fmaf(wi, v.y, u.x)), fmaf(-wr, v.y, fmaf(-wi, v.x, u.y))};
    return r;
}
static int rev(int v, int bits) { int r = 0; for (int b = 0; b < bits; ++b) if (v & (1 << b)) r |= 1 << (bits - 1 - b); return r; }

// in-place DIT on npts (power of 2) points already in bit-reversed slot order, twiddle index step: tw[j * (W / m)]
// with the oracle's special cases (j == 0, 4 j == m)
static void dit_inplace(cplx* x, int npts) {
    for (int m = 2; m <= npts; m <<= 1) {
        int h = m >> 1, tstep = W / m;
        for (int base = 0; base < npts; base += m)
            for (int j = 0; j < h; ++j) {
                cplx u = x[base + j], v = x[base + j + h];
                if (j == 0) { x[base + j] = cadd(u, v); x[base + j + h] = csub(u, v); }
                else if (4 * j == m) { cplx a = {u.x + v.y, u.y - v.x}, b = {u.x - v.y, u.y + v.x}; x[base + j] = a; x[base + j + h] = b; }
                else { x[base + j] = madd(u, twr[j * tstep], twi[j * tstep], v); x[base + j + h] = msub(u, twr[j * tstep], twi[j * tstep], v); }
            }
    }
}

int main(void) {
    lbo_twiddles(W, twr, twi);
    lbo_config cfg; lbo_default_config(&cfg); cfg.sample_rate = 48000; cfg.window = W;
    const int n_windows = 128;
    const int n_samples = (n_windows - 1) * 64 + W + 64;   // + one extra hop: D5 of block 128
    float* pcm = malloc(sizeof(float) * n_samples);
    lbo_synth_clip(0x4C424144, 5, 48000, n_samples, 1, pcm);
    uint32_t idx[33], lo[32], hi[32];
    lbo_band_table(48000, W, W, 32, idx, lo, hi);
    uint32_t kmin = 0xffffffff, kmax = 0;
    for (int b = 0; b < 32; ++b) if (lo[b] < hi[b]) { if (lo[b] < kmin) kmin = lo[b]; if (hi[b] > kmax) kmax = hi[b]; }
    printf("kmin %u kmax %u\n", kmin, kmax);
    const cplx* c = (const cplx*)pcm;    // complex points
    // lane state: P[n][h][16]
    static cplx P[32][2][16], Nw[32][2][16];
    static cplx T[64][32];     // T[row][n]
    int bad = 0;
    for (int step = 0; step <= n_windows; ++step) {
        // ---- phase 1: D5 of block `step` (g = 32 step + n), split over h
        for (int n = 0; n < 32; ++n) {
            cplx d4[2][16];
            for (int h = 0; h < 2; ++h) {
                int g = 32 * step + n + 64 * h;
                cplx x[16];
                for (int t = 0; t < 16; ++t) x[t] = c[g + 128 * rev(t, 4)];
                dit_inplace(x, 16);          // twiddles tw[j * W/m]: m <= 16 -> W_16.. ok (standalone indexes equal the oracle's)
                memcpy(d4[h], x, sizeof(x));
            }
            // stage 5: u = d4[0][k], v = d4[1][k]: lane h=0 keeps u + w v (k), lane h=1 keeps u - w v (k + 16); table twiddle with sign folded
            for (int kk = 0; kk < 16; ++kk) {
                float wr = twr[kk * (W / 32)], wi = twi[kk * (W / 32)];
                Nw[n][0][kk] = madd(d4[0][kk], wr, wi, d4[1][kk]);
                Nw[n][1][kk] = madd(d4[0][kk], -wr, -wi, d4[1][kk]);      // s = -1 folded into the twiddle
            }
        }
        if (step > 0) {
            int win = step - 1;
            // stage 6 + emit
            for (int n = 0; n < 32; ++n)
                for (int h = 0; h < 2; ++h)
                    for (int kk = 0; kk < 16; ++kk) {
                        int k = 16 * h + kk;
                        float wr = twr[k * (W / 64)], wi = twi[k * (W / 64)];
                        T[k][n] = madd(P[n][h][kk], wr, wi, Nw[n][h][kk]);
                        T[k + 32][n] = msub(P[n][h][kk], wr, wi, Nw[n][h][kk]);
                    }
            // ---- phase 2: lane = 2 pp + hh
            static float power[N];
            for (int k = 0; k < N; ++k) power[k] = NAN;
            cplx lo6[64][6], hi6[64][6];
            int rowof[64];
            for (int lane = 0; lane < 64; ++lane) {
                int pp = lane >> 1, hh = lane & 1;
                int a = hh == 0 ? pp : (pp == 0 ? 32 : 64 - pp);
                rowof[lane] = a;
                cplx y[32];
                for (int t = 0; t < 32; ++t) y[t] = T[a][rev(t, 5)];
                // cross fft, general formula everywhere (table twiddles)
                for (int s = 1; s <= 5; ++s) {
                    int half = 1 << (s - 1);
                    for (int b = 0; b < 32; b += 2 * half)
                        for (int jj = 0; jj < half; ++jj) {
                            int ti = (a + 64 * jj) << (6 - s);
                            cplx u = y[b + jj], v = y[b + jj + half];
                            y[b + jj] = madd(u, twr[ti], twi[ti], v);
                            y[b + jj + half] = msub(u, twr[ti], twi[ti], v);
                        }
                }
                for (int q = 0; q < 6; ++q) { lo6[lane][q] = y[q]; hi6[lane][q] = y[26 + q]; }
            }
            for (int lane = 0; lane < 64; ++lane) {
                int pp = lane >> 1, hh = lane & 1, a = rowof[lane];
                for (int q = 0; q < 6; ++q) {
                    int k = a + 64 * q;
                    if (k < (int)kmin || k >= (int)kmax) continue;
                    cplx A = lo6[lane][q], B;
                    if (pp == 0) { if (hh) B = hi6[lane][5 - q]; else { if (q == 0) continue; B = hi6[lane][6 - q]; } }
                    else B = hi6[lane ^ 1][5 - q];
                    float sr = A.x + B.x, si = A.y - B.y, dr = A.x - B.x, di = A.y + B.y;
                    float re = fmaf(twr[k], di, fmaf(twi[k], dr, sr));
                    float im = fmaf(-twr[k], dr, fmaf(twi[k], di, si));
                    float norm = (float)(W / 4);
                    if (re > 0.0f) re /= norm;
                    if (im > 0.0f) im /= norm;
                    power[k] = re * re + im * im;
                }
            }
            float row[32], want[32];
            for (int b = 0; b < 32; ++b) {
                float p = 0.0f;
                for (uint32_t k = lo[b]; k < hi[b]; ++k) { float v = power[k]; if (v == v && isfinite(v)) p += v; else if (v != v) { printf("missing bin %u\n", k); bad++; } }
                row[b] = p / (float)(idx[b + 1] - idx[b]);
            }
            lbo_window_row(pcm + 64 * win, &cfg, want);
            if (memcmp(row, want, sizeof(row)) != 0) {
                if (bad < 5) { printf("window %d differs:", win); for (int b = 0; b < 32; ++b) if (row[b] != want[b]) printf(" b%d %g vs %g", b, row[b], want[b]); printf("\n"); }
                bad++;
            }
        }
        memcpy(P, Nw, sizeof(P));
    }
    printf("%s: %d bad of %d windows\n", bad ? "FAIL" : "OK", bad, n_windows);
    return bad != 0;
}
