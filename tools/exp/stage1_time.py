#!/usr/bin/env python3
"""The configs[1] pass with the library given on the command line (A/B of two builds on one box): stage-1 / stage-2 time
from the library's own events, 10 passes per reading, and a hash of the packed results.
python tools/exp/stage1_time.py [other liblbaudiodetective.so] [config: B | A | C]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lbaudiodetective_amd import _native as N
args = [a for a in sys.argv[1:]]
cfg = "B"
for a in list(args):
    if a in ("A", "B", "C"):
        cfg = a
        args.remove(a)
if args:
    N.LIB_PATH = os.path.abspath(args[0])
import lbaudiodetective_amd as lb
rate, window, n, samples, stereo = {"B": (44100, 1024, 100000, 44100, False), "A": (5512, 2048, 20000, 5512 * 9, False),
                                    "C": (48000, 4096, 10000, 48000, True)}[cfg]
det = lb.Detective().configure(sample_rate=rate, window=window)
clips = lb.synth_clips_device(0x4C424144, 0, n, rate, samples, stereo)
out = det.fingerprint_clips_device(clips)
torch.cuda.synchronize()
for rnd in range(4):
    det.set_stage_timing(True)
    for _ in range(10):
        det.fingerprint_clips_device(clips, out=out)
    s1, s2, launches = det.stage_times()
    det.set_stage_timing(False)
    print(f"{os.path.basename(os.path.dirname(os.path.dirname(N.LIB_PATH)))} {cfg}: stage 1 {s1 / 10:.3f} ms, stage 2 {s2 / 10:.3f} ms per pass; "
          f"bits {hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:12]}")
