import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import lbaudiodetective_amd as lb
variant = int(sys.argv[1])
det = lb.Detective().configure(sample_rate=44100, window=1024)
det.set_kernel_variant(variant)
clips = lb.synth_clips_device(0x4C424144, 0, 20000, 44100, 44100)
out = None
for _ in range(3):
    out = det.fingerprint_clips_device(clips, out=out)
torch.cuda.synchronize()
