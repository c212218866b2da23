"""Per-wave phase timeline of the stage-1 kernel (frame_rows_pruned_kernel).

Build the library with the stamps compiled in, then run on the GPU box:

    make -C lbaudiodetective_amd/csrc OUT=../lib \
         FLAGS_k_rows_pruned="-fno-slp-vectorize -mllvm -amdgpu-atomic-optimizer-strategy=None -DLBAD_EXP_TIMELINE"
    PYTHONPATH=. python tools/exp/timeline.py

With LBAD_EXP_TIMELINE the kernel writes s_memtime deltas (shader-clock ticks) of one lane per wave
into the frame rows instead of the band means, so the results of that build are NOT fingerprints.
s_memtime needs lgkmcnt(0), which drains the LDS queue at every stamp: the instrumented kernel runs
about 10 % slower than the shipped one; compare variants, do not read absolute times off it.
"""
import collections

import numpy as np
import torch

import os, sys
from lbaudiodetective_amd import _native as _N
if len(sys.argv) > 1:
    _N.LIB_PATH = os.path.abspath(sys.argv[1])
import lbaudiodetective_amd as lb

det = lb.Detective().configure(sample_rate=44100.0, window=1024, stride=64)
n = 20000
clips = torch.empty((n, 44100), dtype=torch.float32, device="cuda")
lb.synth_clips_device(0x4C424144, 0, n, 44100, 44100, out=clips)
for _ in range(2):
    out, raw, haar = det.fingerprint_clips_device(clips, taps=True)
torch.cuda.synchronize()
r = raw.cpu().numpy().reshape(n * 5, 4, 4, 8, 32)[:, :, :, 0, :10]   # frame, quarter, wave -> 10 values
r = r.reshape(-1, 4, 10)                                               # quarter frame, wave, values
names = ["wait + barrier", "points, barrier, prefetch issue, FFT*", "FFT*", "pass 1", "pass 2", "bands"]
mid = r[r.shape[0] // 4: 3 * r.shape[0] // 4]
print("quarter frames", r.shape[0], "(* the compiler moves butterflies across the stamp)")
for i, nm in enumerate(names):
    v = mid[:, :, i].ravel()
    print(f"{nm:40s} mean {v.mean():8.0f}  p10 {np.percentile(v, 10):8.0f}  p50 {np.percentile(v, 50):8.0f}  p90 {np.percentile(v, 90):8.0f}")
tot = mid[:, :, :6].sum(axis=2).ravel()
rt = mid[:, :, 9].ravel()
print(f"ticks per quarter frame and wave: mean {tot.mean():.0f}, p50 {np.percentile(tot, 50):.0f}")
print(f"s_memrealtime (100 MHz) per iteration: {rt.mean():.1f} -> shader clock {tot.mean() / (rt.mean() * 10.0):.2f} GHz")
hw = r[:, 0, 7].astype(np.int64)          # HW_ID of wave 0
print("quarter frames by wave slot of the SIMD:", sorted(collections.Counter((hw & 15).tolist()).items()))
print("wave 0 by SIMD:", sorted(collections.Counter(((hw >> 4) & 3).tolist()).items()))
