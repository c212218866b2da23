// plan.cpp -- host-side tables of the fingerprint path (double precision, computed once per
// configuration instead of once per window as LBAudioDetective.m:361-371 does).
#include "internal.hpp"

#include <cmath>

namespace lbad {

OSStatus hip_status(hipError_t e, const char* what, int line) {
    if (e == hipSuccess) return noErr;
    fprintf(stderr, "lbaudiodetective: HIP error %d (%s) at %s, line %d\n", (int)e, hipGetErrorString(e), what, line);
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver)
        return kLBAudioDetectiveDeviceUnavailable;
    return kLBAudioDetectiveDeviceError;
}

bool device_ready() {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0;
}

int current_device() {
    int d = -1;
    return hipGetDevice(&d) == hipSuccess ? d : -1;
}

int device_cu_count() {
    static int cus[kMaxDevices] = {};
    const int d = current_device();
    if (d < 0 || d >= kMaxDevices) return 256;
    if (cus[d] == 0) {
        int n = 0;
        cus[d] = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) == hipSuccess && n > 0 ? n : 256;
    }
    return cus[d];
}

namespace {
// (UInt32) of a Float64: values below zero give 0 (the reference's ARM targets saturate; see
// SURVEY.md Q4 -- negative fractions occur for the first bands at 44.1 kHz / 1024).
uint32_t to_u32(double v) {
    if (!(v > 0.0)) return 0u;
    if (v >= 4294967295.0) return 4294967295u;
    return (uint32_t)v;
}
}  // namespace

void make_band_table(double sample_rate, uint32_t window, uint32_t bands, BandTable& out) {
    out.indices.assign(bands + 1, 0);
    out.lo.assign(bands, 0);
    out.hi.assign(bands, 0);
    // LBAudioDetective.m:362-366
    const double top = sample_rate / 2.0;
    const double bottom = 318.0;
    const double base = std::exp(std::log(top / bottom) / bands);
    const double coef = (double)window / sample_rate * bottom;
    for (uint32_t j = 0; j <= bands; ++j)  // :368-371
        out.indices[j] = to_u32((std::pow(base, (double)j) - 1.0) * coef) + to_u32(coef);
    // :382-383 -- the edges are FFT bin numbers but get converted once more as if they were Hz
    const double bin_hz = sample_rate / window;
    const uint32_t nyq = window / 2;
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0;
    for (uint32_t i = 0; i < bands; ++i) {
        uint32_t l = to_u32((double)(uint32_t)(2u * out.indices[i]) / bin_hz - 1.0);
        uint32_t h = to_u32((double)(uint32_t)(2u * out.indices[i + 1]) / bin_hz - 1.0);
        if (l > nyq) l = nyq;  // bins past the buffer are not read (the reference would overrun)
        if (h > nyq) h = nyq;
        out.lo[i] = l;
        out.hi[i] = h;
        if (l < h) {
            if (l < kmin) kmin = l;
            if (h > kmax) kmax = h;
        }
    }
    if (kmin > kmax) kmin = kmax = 0;
    out.kmin = kmin;
    out.kmax = kmax;
    // the terms' layout (internal.hpp): greedy, every band on the first bank none of the (up to 32) bands before it started on
    out.term_at.assign(bands, 0);
    out.ordered = true;
    uint32_t last_hi = 0;
    for (uint32_t i = 0; i < bands; ++i) {
        if (out.hi[i] <= out.lo[i]) continue;
        if (out.lo[i] < last_hi) out.ordered = false;
        last_hi = out.hi[i];
    }
    if (out.ordered) {
        uint32_t used = 0, at = 0;
        for (uint32_t i = 0; i < bands; ++i) {
            if (used == 0xFFFFFFFFu) used = 0;
            while ((used >> (at & 31u)) & 1u) ++at;
            used |= 1u << (at & 31u);
            out.term_at[i] = at;
            at += out.hi[i] > out.lo[i] ? out.hi[i] - out.lo[i] : 0u;
        }
        out.term_end = at;
    } else {
        for (uint32_t i = 0; i < bands; ++i) out.term_at[i] = out.hi[i] > out.lo[i] ? out.lo[i] - kmin : 0u;
        out.term_end = kmax - kmin;
    }
    out.unread_terms16 = 0;
    if (window == 2048) {
        // the bins of k_rows_full.hip's split pass at 16 lanes per window: lane r, slot 2 r + q, pair u -> bin slot + 64 u and
        // its partner 1024 - bin (slot 0 pairs rows 0 and 32 with themselves: the same re-indexing as in the kernel)
        std::vector<uint8_t> read(nyq + 1, 0);
        for (uint32_t i = 0; i < bands; ++i)
            for (uint32_t k = out.lo[i]; k < out.hi[i]; ++k) read[k] = 1;
        const int L = 16, N = 1024;
        for (int q = 0; q < 2; ++q)
            for (int u = 0; u < L; ++u)
                for (int half = 0; half < 2; ++half) {
                    bool unread = true;
                    for (int r = 0; r < L; ++r) {
                        const int slot = 2 * r + q;
                        int ka = slot + 64 * u;
                        if (slot == 0 && u >= L / 2) ka = 32 + 64 * (u - L / 2);
                        int kb = N - ka;
                        if (slot == 0 && u == 0) kb = N / 2;
                        if (half && kb == ka) continue;
                        if (read[half ? kb : ka]) unread = false;
                    }
                    if (unread) out.unread_terms16 |= 1ull << ((q * L + u) * 2 + half);
                }
    }
}

void make_band_bounds(double sample_rate, uint32_t window, uint32_t n_frames, const BandTable& table, uint32_t* lo,
                      uint32_t* hi) {
    const double bin_hz = sample_rate / (double)n_frames;   // :382-383 with inNumberFrames = n_frames
    const uint32_t nyq = window / 2;
    const uint32_t bands = (uint32_t)table.lo.size();
    for (uint32_t i = 0; i < bands; ++i) {
        uint32_t l = to_u32((double)(uint32_t)(2u * table.indices[i]) / bin_hz - 1.0);
        uint32_t h = to_u32((double)(uint32_t)(2u * table.indices[i + 1]) / bin_hz - 1.0);
        lo[i] = l > nyq ? nyq : l;
        hi[i] = h > nyq ? nyq : h;
    }
}

void make_twiddles(uint32_t W, std::vector<float>& re, std::vector<float>& im) {
    const uint32_t half = W / 2, quarter = W / 4, eighth = W / 8;
    re.assign(half, 0.0f);
    im.assign(half, 0.0f);
    // first octant from libm in double, the rest by symmetry so that e.g. cos(pi/4) == sin(pi/4)
    // and cos(pi/2) == 0 hold exactly in float32
    for (uint32_t k = 0; k < half; ++k) {
        uint32_t f = k;
        bool mirror = false, transpose = false;
        if (f > quarter) { f = half - f; mirror = true; }
        if (f > eighth) { f = quarter - f; transpose = true; }
        const double a = (2.0 * M_PI * (double)f) / (double)W;
        float c = (float)std::cos(a), s = (float)std::sin(a);
        if (f == eighth) s = c;
        if (transpose) std::swap(c, s);
        if (mirror) c = -c;
        if (k == quarter) c = 0.0f;
        re[k] = c;
        im[k] = -s;
    }
}

// Which bands can be non-zero at all, and what that leaves of the row transform (LBAudioDetectiveFrame.m:134-153 on a
// row of 32): a band whose bin range is empty is 0 / divisor = +0.0 in every window (LBAudioDetective.m:386-404).  The
// ordered positions of a row's Haar output are [0] the average, [1] the level-5 detail, [2..3] level 4, [4..7] level 3,
// [8..15] level 2, [16..31] level 1; an output is structurally zero when both its inputs are.
void plan_sparse(Plan& plan) {
    Plan::Sparse sp;
    plan.sparse = sp;
    if (plan.bands != 32) return;
    bool live[32];
    uint32_t n_left = 0;
    for (uint32_t b = 0; b < 32; ++b) {
        live[b] = plan.table.lo[b] < plan.table.hi[b];
        if (b < 16 && live[b]) { ++n_left; sp.left = b; }
        // an empty band is 0 / divisor (LBAudioDetective.m:404): +0.0 -- unless the divisor is 0 as well (two equal band
        // edges, e.g. 48 kHz / 1024), which makes it NaN in every window: such a table keeps the general kernels
        if (!live[b] && plan.table.indices[b + 1] == plan.table.indices[b]) return;
    }
    if (n_left > 1) return;
    // propagate "can be non-zero" through the five levels
    bool cur[32], out[32] = {};
    for (int i = 0; i < 32; ++i) cur[i] = live[i];
    for (int n = 32; n > 1; n >>= 1) {
        bool next[16];
        for (int i = 0; i < n / 2; ++i) {
            const bool any = cur[2 * i] || cur[2 * i + 1];
            next[i] = any;
            out[n / 2 + i] = any;          // the level's details sit at [n / 2, n)
        }
        for (int i = 0; i < n / 2; ++i) cur[i] = next[i];
    }
    out[0] = cur[0];
    for (uint32_t c = 0; c < 32; ++c)
        if (out[c]) sp.cols[sp.n_cols++] = (uint8_t)c;
    if (sp.n_cols == 0 || sp.n_cols > 21) return;   // nothing is ever read / more columns than the sparse stage 2 has slots for (k_haar_select32.hip: kSparseCols)
    // what a compact frame's row holds: the live bands of the right half in ascending order, then the left half's live band
    for (uint32_t j = 0; j < 16; ++j) {
        sp.pos_right[j] = 0xFF;
        if (live[16 + j]) { sp.pos_right[j] = (uint8_t)sp.n_stored; sp.stored[sp.n_stored++] = (uint8_t)(16 + j); }
    }
    if (sp.left < 32) { sp.pos_left = (uint8_t)sp.n_stored; sp.stored[sp.n_stored++] = (uint8_t)sp.left; }
    if (sp.n_stored == 0) return;
    sp.ok = true;
    plan.sparse = sp;
}

}  // namespace lbad
