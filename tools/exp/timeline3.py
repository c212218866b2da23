"""Phase timeline of the SPARSE stage 2 (haar_select32_kernel<32, true>; build with -DLBAD_EXP_TIMELINE, LBAD_LIB=...):
shader-clock ticks per phase on configs[1]'s own stage-1 output."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lbaudiodetective_amd import _native as _N
if os.environ.get("LBAD_LIB"):
    _N.LIB_PATH = os.path.abspath(os.environ["LBAD_LIB"])
import lbaudiodetective_amd as lb
det = lb.Detective().configure(sample_rate=44100.0, window=1024, stride=64)
n = 6250
clips = lb.synth_clips_device(0x4C424144, 0, n, 44100, 44100)
_, raw, _ = det.fingerprint_clips_device(clips, taps=True)
frames = raw.reshape(-1, 128, 32).contiguous()
for compact in (True, False):
    for _ in range(3):
        out, haar = lb.frames_to_subfingerprints_device(det, frames, want_haar=True, compact=compact)
    torch.cuda.synchronize()
    h = haar.cpu().numpy().reshape(frames.shape[0], 4096)[:, :7]
    names = ["load+row pass", "column pass", "threshold search", "gather", "rank+emit"]
    mid = h[h.shape[0] // 4: 3 * h.shape[0] // 4]
    print("sparse form" if compact else "general form")
    for i, nm in enumerate(names):
        v = mid[:, i]
        print(f"  {nm:20s} mean {v.mean():8.0f} p10 {np.percentile(v,10):8.0f} p50 {np.percentile(v,50):8.0f} p90 {np.percentile(v,90):8.0f}")
    print("  total", mid[:, :5].sum(axis=1).mean(), "bisection steps mean", mid[:, 5].mean(), "candidates mean", mid[:, 6].mean())
