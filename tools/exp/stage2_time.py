#!/usr/bin/env python3
"""Stage 2 alone (haar_select32_kernel) on real stage-1 output: 31 250 frames of configuration B (one 512 MiB chunk of
the bench pass), and on random 16- / 64-band frames.  `python tools/exp/stage2_time.py [other liblbaudiodetective.so]`
prints the average kernel time and a hash of the packed bits, so two builds can be A/B-ed on one box."""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import lbaudiodetective_amd as lb
from lbaudiodetective_amd import _native as N

if len(sys.argv) > 1:
    N.LIB_PATH = os.path.abspath(sys.argv[1])
lb.lib()
out = {"lib": N.LIB_PATH}


def run(det, frames, reps=200):
    packed = lb.frames_to_subfingerprints_device(det, frames)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lb.frames_to_subfingerprints_device(det, frames)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, hashlib.sha256(packed.cpu().numpy().tobytes()).hexdigest()[:16]


det = lb.Detective().configure(sample_rate=44100, window=1024)
clips = lb.synth_clips_device(0x4C424144, 0, 6250, 44100, 44100)
_, raw, _ = det.fingerprint_clips_device(clips, taps=True)
torch.cuda.synchronize()
frames = raw.reshape(-1, 128, 32).contiguous()
del clips
for rnd in range(4):
    us, h = run(det, frames)
    out[f"B_32_bands_{frames.shape[0]}_frames_round{rnd}"] = {"us": round(us, 1), "bits": h}
del frames, raw
torch.manual_seed(5)
for bands, n in ((16, 31250), (64, 15625)):
    d2 = lb.Detective().configure(sample_rate=44100, window=1024, bands=bands)
    fr = (torch.rand(n, 128, bands, device="cuda") ** 4) * 50.0 + 1e-6
    fr[:, :, ::3] = 0.0
    us, h = run(d2, fr)
    out[f"{bands}_bands_{n}_frames"] = {"us": round(us, 1), "bits": h}
print(json.dumps(out, indent=1))
