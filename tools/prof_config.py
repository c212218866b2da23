#!/usr/bin/env python3
"""Driver for rocprofv3 passes over ONE processing configuration of BASELINE.json:
    python3 tools/prof_config.py <A|B|C> [n_clips] [variant] [reps] [waves cache [scratch MB]]   (waves -1: leave the tuning)
A = defaults 5512 Hz / 2048 (9 s clips), B = 44.1 kHz / 1024 (1 s), C = 48 kHz / 4096 stereo-summed (1 s).
A device-to-device copy of the clip buffer follows (known byte count: calibrates FETCH_SIZE / WRITE_SIZE)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lbaudiodetective_amd as lb

CFG = {"A": (5512, 2048, 5512 * 9, False, 20000), "B": (44100, 1024, 44100, False, 20000), "C": (48000, 4096, 48000, True, 10000)}
name = sys.argv[1] if len(sys.argv) > 1 else "C"
rate, window, samples, stereo, n_default = CFG[name]
n = int(sys.argv[2]) if len(sys.argv) > 2 else n_default
variant = int(sys.argv[3]) if len(sys.argv) > 3 else 0
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
det = lb.Detective().configure(sample_rate=rate, window=window)
det.set_kernel_variant(variant)
if len(sys.argv) > 6 and int(sys.argv[5]) >= 0:
    det.set_kernel_tuning(int(sys.argv[5]), bool(int(sys.argv[6])))
if len(sys.argv) > 7:
    det.set_scratch_limit(int(sys.argv[7]) << 20)      # MB of inter-stage frame rows (chunks the batch)
clips = lb.synth_clips_device(0x4C424144, 0, n, rate, samples, stereo)
out = None
for _ in range(reps):
    out = det.fingerprint_clips_device(clips, out=out)
torch.cuda.synchronize()
dst = torch.empty_like(clips)
for _ in range(2):
    dst.copy_(clips)
torch.cuda.synchronize()
print("done", name, n, variant, "clip bytes", clips.numel() * 4, "windows", n * out.shape[1] * 128)
