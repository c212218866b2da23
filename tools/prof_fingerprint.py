#!/usr/bin/env python3
"""Small driver for rocprofv3 counter passes: a few launches of the fingerprint path on N clips."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lbaudiodetective_amd as lb

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
det = lb.Detective().configure(sample_rate=44100, window=1024)
det.set_kernel_variant(variant)
clips = lb.synth_clips_device(0x4C424144, 0, n, 44100, 44100)
out = None
for _ in range(reps):
    out = det.fingerprint_clips_device(clips, out=out)
torch.cuda.synchronize()
print("done", n, variant)
