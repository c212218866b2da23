"""Does keeping the inter-stage frame rows inside the 256 MB Infinity Cache pay?  The configs[1] pass with the
scratch limit (LBAudioDetectiveSetScratchLimit) at several sizes: pass time from events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import lbaudiodetective_amd as lb
n = 100000
clips = lb.synth_clips_device(0x4C424144, 0, n, 44100, 44100)
for mb in (16384, 2048, 1024, 512, 384, 256, 192, 128, 64, 32):
    det = lb.Detective().configure(sample_rate=44100, window=1024)
    det.set_scratch_limit(mb << 20)
    out = det.fingerprint_clips_device(clips)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        det.fingerprint_clips_device(clips, out=out)
    e1.record(); torch.cuda.synchronize()
    det.set_stage_timing(True)
    det.fingerprint_clips_device(clips, out=out)
    s1, s2, ln = det.stage_times()
    print(f"scratch {mb:6d} MB: pass {e0.elapsed_time(e1) / 20:.3f} ms  (stage 1 {s1:.2f} + stage 2 {s2:.2f} ms in {ln} chunk(s))")
    if len(sys.argv) > 1 and int(sys.argv[1]) == mb:
        break
