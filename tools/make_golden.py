#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_vectors.npz from the CPU oracle (oracle/lbad_oracle.c).

These vectors are produced by OUR restatement, not by the upstream binary (its FFT is Apple's
vDSP, which does not exist on Linux): they guard the oracle against drift and give the GPU
parity tests fixed inputs.  PCM is stored as int16 (sample = value / 32768).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

SEED = 0x4C424144
out = {}

# B: 44.1 kHz / 1024 (BASELINE configs 2-4) -- two 1 s clips
cfg = O.Config(44100, 1024)
pcm = O.synth_clips(SEED, 0, 2, 44100, 44100)
out["B_pcm_i16"] = np.round(pcm * 32768).astype(np.int16)
bits, raw, haar = O.fingerprint_pcm(pcm[0], cfg, taps=True)
out["B_bits"] = np.stack([bits, O.fingerprint_pcm(pcm[1], cfg)])
out["B_raw_frame0"] = raw[0]
out["B_haar_frame0"] = haar[0]

# A: defaults 5512 Hz / 2048 (BASELINE config 1) -- one 3 s clip
cfg = O.Config()
pcm = O.synth_clip(SEED, 7, 5512, 16536)
out["A_pcm_i16"] = np.round(pcm * 32768).astype(np.int16)
bits, raw, haar = O.fingerprint_pcm(pcm, cfg, taps=True)
out["A_bits"], out["A_raw_frame0"], out["A_haar_frame0"] = bits, raw[0], haar[0]

# C: 48 kHz / 4096, stereo summed (BASELINE config 5) -- int32 numerator over 65536
cfg = O.Config(48000, 4096)
pcm = O.synth_clip(SEED, 11, 48000, 48000, stereo_sum=True)
out["C_pcm_i32"] = np.round(pcm * 65536).astype(np.int32)
out["C_bits"] = O.fingerprint_pcm(pcm, cfg)

# compare: (n1, n2, range) cases over random / pathological Boolean sets
rng = np.random.default_rng(20131001)
cases, a_all, b_all, exp = [], [], [], []
for (n1, n2) in [(1, 1), (5, 5), (8, 3), (3, 8), (21, 48), (2, 7)]:
    for rg in (200, 100, 37, 2, 1, 1000):
        def draw(n):
            pos = rng.integers(0, 2, (n, 100))
            zero = rng.random((n, 100)) < 0.05
            f = np.zeros((n, 200), np.uint8)
            f[:, 0::2] = pos & ~zero
            f[:, 1::2] = (1 - pos) & ~zero
            return f
        a, b = draw(n1), draw(n2)
        if n1 == n2 == 5 and rg == 100:
            b = a.copy(); b[:, :14] ^= 1
        cases.append((n1, n2, rg))
        a_all.append(a.reshape(-1)); b_all.append(b.reshape(-1))
        exp.append(np.float32(O.compare_fp(a, b, rg)))
out["cmp_cases"] = np.array(cases, np.int32)
out["cmp_a"] = np.concatenate(a_all)
out["cmp_b"] = np.concatenate(b_all)
out["cmp_expected_bits"] = np.array(exp, np.float32).view(np.uint32)

path = os.path.join(ROOT, "tests", "golden", "oracle_vectors.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path), "bytes")
