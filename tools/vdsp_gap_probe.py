#!/usr/bin/env python3
"""How much can the UNKNOWN rounding order of Apple's vDSP_fft_zrip change a fingerprint?

The reference transforms every window with vDSP (LBAudioDetective.m:353-355), closed source and absent here;
the oracle (and the HIP kernels, bit for bit) use one fixed float32 evaluation: radix-2 decimation in time,
nested-fmaf butterflies, folded split pass.  This probe runs the REST of the oracle pipeline (band means, Haar,
ranked signs -- lbo_spectra_to_rows / lbo_rows_to_subfingerprints) behind alternative FFT evaluations of the same
mathematical transform and counts what changes:

  f64          numpy's float64 rfft, rounded to float32 once (the exact transform, as far as float32 can tell)
  pocket32     scipy.fft.rfft on float32 input: pocketfft's float32 mixed-radix (4/2) real transform, an
               independent library implementation -- the closest stand-in for "another vendor's FFT"
  dit_nofma    the oracle's own DIT order with separately rounded multiplies and adds (no FMA)
  dif_nofma    radix-2 decimation in FREQUENCY, no FMA
  radix4_nofma radix-4 DIT (radix-2 last stage where log2 N is odd), no FMA
  tw_float     DIT, no FMA, twiddles from float32 cosf/sinf instead of a rounded double-precision table

Inputs: the synthetic clips of BASELINE's three processing configurations and the sixty upstream bird
fixtures through upstream's file loop (hop mode 1, tail mode 1).  Reported per variant: bits flipped,
sub-fingerprints touched, and the largest shift of a match value (birds: the 10 x 10 matrices of the five
upstream tests; synthetic: variant fingerprint against canonical fingerprint of the same clip).

    python tools/vdsp_gap_probe.py [--quick] [--json profiles/r02_vdsp_gap.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.fft

from oracle import oracle as O

BIRDS = ["BlackBird", "BlueTit", "Chaffinch", "Sparrow", "GreatTit", "Crow", "Wren", "Chiffchaff", "Kestrel", "Pigeon"]
SUFFIXES = ["_eql", "_dif", "_blu1", "_blu2", "_rec"]
DIR = os.path.join(ROOT, "tests", "golden", "birds")
SEED = 0x4C424144


# ---- FFT variants: [n, W] float32 windows -> [n, W] float32 packed 2 x DFT (vDSP_fft_zrip + ztoc layout) ------
def pack(X):
    """complex [n, W/2 + 1] DFT -> packed float32 (x2, DC / Nyquist in elements 0 / 1)."""
    n, half = X.shape[0], X.shape[1] - 1
    out = np.empty((n, 2 * half), np.float32)
    out[:, 0] = (2 * X[:, 0].real).astype(np.float32)
    out[:, 1] = (2 * X[:, half].real).astype(np.float32)
    out[:, 2::2] = (2 * X[:, 1:half].real).astype(np.float32)
    out[:, 3::2] = (2 * X[:, 1:half].imag).astype(np.float32)
    return out


def fft_f64(w):
    return pack(np.fft.rfft(w.astype(np.float64), axis=1))


def fft_pocket32(w):
    X = scipy.fft.rfft(np.ascontiguousarray(w, np.float32), axis=1)
    assert X.dtype == np.complex64
    return pack(X)


def _twiddles(W, float_trig=False):
    if float_trig:
        k = np.arange(W // 2, dtype=np.float32)
        a = (np.float32(2.0 * np.pi) * k) / np.float32(W)
        return np.cos(a).astype(np.float32), (-np.sin(a)).astype(np.float32)
    return O.twiddles(W)


def _split_pass(zr, zi, twr, twi):
    """Y[k] = (A + conj B) - i w (A - conj B), x2 folded in (the oracle's split pass), no FMA."""
    n, N = zr.shape
    out = np.empty((n, 2 * N), np.float32)
    s, d = zr[:, 0] + zi[:, 0], zr[:, 0] - zi[:, 0]
    out[:, 0], out[:, 1] = s + s, d + d
    k = np.arange(1, N)
    ar, ai, br, bi = zr[:, k], zi[:, k], zr[:, N - k], zi[:, N - k]
    sr, si, dr, di = ar + br, ai - bi, ar - br, ai + bi
    wr, wi = twr[k], twi[k]
    out[:, 2::2] = (sr + wi * dr) + wr * di
    out[:, 3::2] = (si + wi * di) - wr * dr
    return out


def _bitrev(N):
    bits = N.bit_length() - 1
    return np.array([int(format(i, f"0{bits}b")[::-1], 2) for i in range(N)])


def fft_dit_nofma(w, float_trig=False):
    n, W = w.shape
    N = W // 2
    twr, twi = _twiddles(W, float_trig)
    rev = _bitrev(N)
    zr, zi = w[:, 0::2][:, rev].copy(), w[:, 1::2][:, rev].copy()
    m = 2
    while m <= N:
        h, step = m // 2, W // m
        wr, wi = twr[np.arange(h) * step], twi[np.arange(h) * step]
        a = zr.reshape(n, N // m, 2, h)
        b = zi.reshape(n, N // m, 2, h)
        ur, ui, vr, vi = a[:, :, 0], b[:, :, 0], a[:, :, 1], b[:, :, 1]
        tr, ti = wr * vr - wi * vi, wr * vi + wi * vr
        a[:, :, 0], b[:, :, 0], a[:, :, 1], b[:, :, 1] = ur + tr, ui + ti, ur - tr, ui - ti
        m *= 2
    return _split_pass(zr, zi, twr, twi)


def fft_tw_float(w):
    return fft_dit_nofma(w, float_trig=True)


def fft_dif_nofma(w):
    n, W = w.shape
    N = W // 2
    twr, twi = _twiddles(W)
    zr, zi = w[:, 0::2].copy(), w[:, 1::2].copy()
    m = N
    while m >= 2:
        h, step = m // 2, W // m
        wr, wi = twr[np.arange(h) * step], twi[np.arange(h) * step]
        a = zr.reshape(n, N // m, 2, h)
        b = zi.reshape(n, N // m, 2, h)
        ur, ui, vr, vi = a[:, :, 0].copy(), b[:, :, 0].copy(), a[:, :, 1].copy(), b[:, :, 1].copy()
        dr, di = ur - vr, ui - vi
        a[:, :, 0], b[:, :, 0] = ur + vr, ui + vi
        a[:, :, 1], b[:, :, 1] = wr * dr - wi * di, wr * di + wi * dr
        m //= 2
    rev = _bitrev(N)
    return _split_pass(zr[:, rev], zi[:, rev], twr, twi)


def fft_radix4_nofma(w):
    """Radix-4 DIT over the N = W/2 complex points (one radix-2 split where log2 of the length is odd)."""
    n, W = w.shape
    twr, twi = _twiddles(W)
    half = W // 2
    z = (w[:, 0::2] + 1j * w[:, 1::2]).astype(np.complex64)

    def tw_at(idx):            # exp(-2 pi i idx / W) for idx in [0, W) from the half table
        sign = np.where(idx >= half, -1.0, 1.0).astype(np.float32)
        return twr[idx % half] * sign, twi[idx % half] * sign

    def cmul(a, idx):          # complex64 multiply, float32 roundings, no FMA
        wr_, wi_ = tw_at(idx)
        return ((a.real * wr_ - a.imag * wi_) + 1j * (a.real * wi_ + a.imag * wr_)).astype(np.complex64)

    def rec(x):                # x: [n, L] -> DFT_L along axis 1
        L = x.shape[1]
        if L == 1:
            return x
        if (L.bit_length() - 1) % 2 == 1:     # odd power of two: one radix-2 split
            e, o = rec(x[:, 0::2]), rec(x[:, 1::2])
            t = cmul(o, np.arange(L // 2) * (W // L))
            return np.concatenate([e + t, e - t], axis=1)
        q = [rec(x[:, r::4]) for r in range(4)]
        k = np.arange(L // 4) * (W // L)
        t1, t2, t3 = cmul(q[1], k), cmul(q[2], 2 * k), cmul(q[3], 3 * k)
        a0, a1 = q[0] + t2, q[0] - t2
        b0, b1 = t1 + t3, t1 - t3
        jb1 = (b1.imag - 1j * b1.real).astype(np.complex64)     # -i * b1
        return np.concatenate([a0 + b0, a1 + jb1, a0 - b0, a1 - jb1], axis=1)

    Z = rec(z)
    return _split_pass(np.ascontiguousarray(Z.real), np.ascontiguousarray(Z.imag), twr, twi)


VARIANTS = {"f64": fft_f64, "pocket32": fft_pocket32, "dit_nofma": fft_dit_nofma, "dif_nofma": fft_dif_nofma,
            "radix4_nofma": fft_radix4_nofma, "tw_float": fft_tw_float}


# ---- the pipeline behind a given FFT ------------------------------------------------------------------------
def windows_of(pcm, cfg, hop, n_windows):
    idx = (np.arange(n_windows) * hop)[:, None] + np.arange(cfg.window)[None, :]
    return pcm[idx]


def fingerprint_with(fft, pcm, cfg, hop=None, n_frames=None, first_short=None):
    """pcm -> sub-fingerprints with `fft` (None = the canonical transform) in place of rfft_exec.  hop /
    n_frames / first_short describe upstream's file loop (rows of short windows are zero, tail mode 1)."""
    hop = hop or cfg.stride
    if n_frames is None:
        n_frames = O.subfingerprint_count(pcm.size, cfg.window, cfg.stride)
    rows_n = n_frames * 128
    full = rows_n if first_short is None else min(rows_n, first_short)
    rows = np.zeros((rows_n, cfg.bands), np.float32)
    for a in range(0, full, 2048):
        b = min(full, a + 2048)
        w = windows_of(pcm[a * hop:], cfg, hop, b - a)
        spec = O.rfft_packed_batch(w) if fft is None else fft(w)
        rows[a:b] = O.spectra_to_rows(spec, cfg)
    return O.rows_to_subfingerprints(rows, cfg)


def synthetic_inputs(quick):
    out = []
    scale = {False: (40, 400, 60), True: (6, 40, 8), "tiny": (2, 12, 2)}[quick]
    for name, rate, window, secs, count, stereo in [("A_5512_2048", 5512, 2048, 9, scale[0], False),
                                                     ("B_44100_1024", 44100, 1024, 1, scale[1], False),
                                                     ("C_48000_4096", 48000, 4096, 1, scale[2], True)]:
        cfg = O.Config(rate, window)
        for c in range(count):
            out.append((name, cfg, O.synth_clip(SEED, c, rate, rate * secs, stereo), None))
    return out


def bird_inputs(quick):
    import lbaudiodetective_amd as lb
    cfg = O.Config()
    names = BIRDS + [b + s for s in SUFFIXES for b in BIRDS]
    if quick == "tiny":
        names = ["Crow", "Crow_eql", "Crow_blu1"]
    elif quick:
        names = BIRDS[:3] + [b + "_eql" for b in BIRDS[:3]]
    out = []
    for n in names:
        p = os.path.join(DIR, n + ".caf")
        x, rate = lb.read_audio_url(p)
        y, _ = lb.read_audio_url(p, cfg.sample_rate)
        hop = max(1, int(round(cfg.stride * cfg.sample_rate / rate)))
        frames = ((x.size - cfg.window) // cfg.stride) // 128
        first_short = (y.size - cfg.window) // hop + 1
        need = frames * 128 * hop + cfg.window
        y = np.concatenate([y, np.zeros(max(0, need - y.size), np.float32)])
        out.append((n, cfg, y, (hop, frames, first_short)))
    return out


def run(quick=False, variants=None):
    variants = variants or list(VARIANTS)
    syn = synthetic_inputs(quick)
    birds = bird_inputs(quick)
    report = {"inputs": {"synthetic_clips": len(syn), "bird_files": len(birds)}, "variants": {}}
    base_syn = [fingerprint_with(None, pcm, cfg) for _, cfg, pcm, _ in syn]
    base_bird = {n: fingerprint_with(None, pcm, cfg, *loop) for n, cfg, pcm, loop in birds}
    # the stage split reproduces the monolithic oracle
    n0, cfg0, pcm0, _ = syn[0]
    assert np.array_equal(base_syn[0], O.fingerprint_pcm(pcm0, cfg0))

    def matrix(fps, suffix):
        have = [b for b in BIRDS if b in fps and b + suffix in fps]
        return np.array([[O.compare_fp(fps[a], fps[b + suffix], 200) for b in have] for a in have])

    for v in variants:
        fft = VARIANTS[v]
        r = {"synthetic": {}, "birds": {}}
        per_cfg = {}
        for (name, cfg, pcm, _), base in zip(syn, base_syn):
            got = fingerprint_with(fft, pcm, cfg)
            d = per_cfg.setdefault(name, {"bits": 0, "flipped": 0, "subfingerprints": 0, "subfingerprints_touched": 0,
                                          "clips": 0, "clips_touched": 0, "min_self_match": 1.0})
            diff = got != base
            d["bits"] += int(base.size); d["flipped"] += int(diff.sum())
            d["subfingerprints"] += base.shape[0]; d["subfingerprints_touched"] += int(diff.any(axis=1).sum())
            d["clips"] += 1; d["clips_touched"] += int(diff.any())
            d["min_self_match"] = min(d["min_self_match"], float(O.compare_fp(got, base, 200)))
        r["synthetic"] = per_cfg
        fps = {n: fingerprint_with(fft, pcm, cfg, *loop) for n, cfg, pcm, loop in birds}
        bits = sum(base_bird[n].size for n in fps)
        flipped = sum(int((fps[n] != base_bird[n]).sum()) for n in fps)
        touched = sum(int((fps[n] != base_bird[n]).any(axis=1).sum()) for n in fps)
        shift = 0.0
        for s in SUFFIXES:
            m0, m1 = matrix(base_bird, s), matrix(fps, s)
            if m0.size:
                shift = max(shift, float(np.abs(m1 - m0).max()))
        r["birds"] = {"files": len(fps), "bits": bits, "flipped": flipped,
                      "subfingerprints": sum(base_bird[n].shape[0] for n in fps), "subfingerprints_touched": touched,
                      "max_match_shift": shift}
        tb = bits + sum(d["bits"] for d in per_cfg.values())
        tf = flipped + sum(d["flipped"] for d in per_cfg.values())
        r["total"] = {"bits": tb, "flipped": tf, "flip_rate": tf / tb}
        report["variants"][v] = r
    return report


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--json")
    args = ap.parse_args()
    rep = run(args.quick)
    for v, r in rep["variants"].items():
        print(f"{v:13s} flipped {r['total']['flipped']:6d} of {r['total']['bits']} bits ({r['total']['flip_rate']:.2e}); "
              f"birds: {r['birds']['subfingerprints_touched']} of {r['birds']['subfingerprints']} sub-fingerprints touched, "
              f"max match shift {r['birds']['max_match_shift']:.2e}")
        for name, d in r["synthetic"].items():
            print(f"      {name}: {d['flipped']} of {d['bits']} bits, {d['clips_touched']} of {d['clips']} clips, "
                  f"min match against canonical {d['min_self_match']:.6f}")
    if args.json:
        json.dump(rep, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
