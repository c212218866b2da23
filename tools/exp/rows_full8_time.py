"""Stage-1 time of k_rows_full.hip at EIGHT lanes per window (1024-sample windows, a table the pruned kernel does not take):
22 050 Hz / 1024, 20 000 clips of 1 s; LBAD_LIB selects the build (tools/exp/build_variants.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import lbaudiodetective_amd as lb
rate, window, n = 22050, 1024, 20000
det = lb.Detective().configure(sample_rate=rate, window=window)
det.set_stage_timing(True)
clips = lb.synth_clips_device(0x4C424144, 0, n, rate, rate, False)
out = None
best = 1e9
for _ in range(6):
    out = det.fingerprint_clips_device(clips, out=out)
    torch.cuda.synchronize()
    s1, s2 = det.stage_times()[:2]
    best = min(best, s1)
print("lib", os.environ.get("LBAD_LIB", "in-tree"), "stage 1 best of 6: %.4f ms for %d clips" % (best, n))
