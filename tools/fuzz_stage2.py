#!/usr/bin/env python3
"""Randomized GPU-vs-oracle sweep of stage 2 ALONE on 128 x {16, 32, 64} frames (the register kernel
k_haar_select32.hip and, with variant 1, the generic one): magnitudes from 2^-140 to 2^120, mixtures of scales inside
one frame, exact zeros, near-equal neighbours (sums that cancel level after level), plateaus of equal values, NaN and
inf -- what the three tiers of the division shortcut, the threshold search and the tie rule have to get right.
tools/fuzz_stage2.py [frames per shape] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lbaudiodetective_amd as lb
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
t0 = time.time()


def make(cols):
    with np.errstate(all="ignore"):
        base = np.abs(rng.standard_normal((128, cols))).astype(np.float32)
        kind = int(rng.integers(0, 10))
        scale = np.float32(2.0 ** float(rng.integers(-140, 121)))
        if kind == 0:
            m = base * scale
        elif kind == 1:                                             # every row (or column) its own scale
            e = rng.integers(-120, 100, (128, 1) if rng.random() < 0.5 else (1, cols))
            m = base * np.exp2(e).astype(np.float32)
        elif kind == 2:                                             # sparse
            m = base * scale
            m[rng.random(m.shape) < rng.random()] = 0
        elif kind == 3:                                             # near-equal values: differences of a few ulps
            m = scale * (np.float32(1.0) + rng.integers(-4, 5, (128, cols)).astype(np.float32) * np.float32(2.0 ** -23))
        elif kind == 4:                                             # plateaus
            m = np.round(base * 3) * scale
        elif kind == 5:                                             # signs mixed
            m = rng.standard_normal((128, cols)).astype(np.float32) * scale
        elif kind == 6:                                             # a few tiny values among ordinary ones
            m = base * np.float32(100.0)
            m[rng.random(m.shape) < 0.01] *= np.float32(2.0 ** float(rng.integers(-149, -40)))
        elif kind == 7:
            m = base * scale
            m[rng.integers(0, 128), rng.integers(0, cols)] = np.float32(np.nan)
        elif kind == 8:
            m = base * scale
            m[rng.integers(0, 128), rng.integers(0, cols)] = np.float32(np.inf) * (1 if rng.random() < 0.5 else -1)
        else:                                                       # the band pattern of 44.1 kHz / 1024: most columns empty
            m = np.zeros((128, cols), np.float32)
            live = rng.random(cols) < 0.45
            m[:, live] = (base * scale)[:, live]
        return np.ascontiguousarray(m, np.float32)


# "sparse": the sparse form of stage 2 (round 4) on frames masked to the band structure of 44.1 kHz / 1024 -- bands 13, 16,
# 18, 20..31 live -- with rows and whole bands knocked out at random on top, so that frames with fewer
# non-zero coefficients than are kept come up often
LIVE = np.zeros(32, bool)
LIVE[[13, 16, 18] + list(range(20, 32))] = True


def make_sparse():
    with np.errstate(all="ignore"):
        m = np.where(LIVE, make(32), np.float32(0)).astype(np.float32)
        r = rng.random()
        if r < 0.25:
            m[rng.random(128) < rng.random()] = 0                   # rows of digital silence
        elif r < 0.4:
            m[:, rng.random(32) < 0.7] = 0                          # most live bands silent too
        elif r < 0.5:
            keep = rng.integers(0, 128, int(rng.integers(1, 4)))
            z = np.zeros_like(m); z[keep] = m[keep]; m = z          # one to three rows
        return np.ascontiguousarray(m, np.float32)


for cols, keep_len, mode in ((32, 200, "dense"), (16, 200, "dense"), (64, 256, "dense"), (32, 64, "dense"), (32, 200, "sparse"),
                            (32, 31, "sparse")):
    for variant in (0, 1) if cols == 32 and keep_len == 200 and mode == "dense" else (0,):
        det = lb.Detective().configure(sample_rate=44100, window=1024, bands=cols, subfp_len=keep_len)
        det.set_kernel_variant(variant)
        todo = n if variant == 0 else max(1, n // 4)
        for b0 in range(0, todo, 256):
            frames = np.stack([(make_sparse() if mode == "sparse" else make(cols)) for _ in range(min(256, todo - b0))])
            packed, haar = lb.frames_to_subfingerprints_device(det, torch.from_numpy(frames).cuda(), want_haar=True,
                                                               compact=mode == "sparse")
            torch.cuda.synchronize()
            got_bits = lb.unpack_packed(packed.cpu().numpy(), keep_len)
            got_haar = haar.cpu().numpy()
            for i in range(frames.shape[0]):
                with np.errstate(all="ignore"):
                    want = O.haar_2d(frames[i])
                ok = np.array_equal(got_haar[i], want, equal_nan=True)
                if ok and not np.isnan(want).any():                 # NaN payloads / signs are not comparable across CPU and GPU
                    ok = np.array_equal(got_bits[i], O.extract(want, keep_len)[:keep_len])
                if not ok:
                    bad += 1
                    print("STAGE-2 MISMATCH", cols, keep_len, variant, mode, b0 + i, flush=True)
print(f"{n} frames per shape, {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
