"""Phase timeline of haar_select32_kernel (build with -DLBAD_EXP_TIMELINE): shader-clock ticks per phase."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lbaudiodetective_amd import _native as _N
if os.environ.get("LBAD_LIB"):
    _N.LIB_PATH = os.path.abspath(os.environ["LBAD_LIB"])
import lbaudiodetective_amd as lb
det = lb.Detective().configure(sample_rate=44100.0, window=1024, stride=64)
n = 20000
clips = torch.empty((n, 44100), dtype=torch.float32, device="cuda")
lb.synth_clips_device(0x4C424144, 0, n, 44100, 44100, out=clips)
for _ in range(2):
    out, raw, haar = det.fingerprint_clips_device(clips, taps=True)
torch.cuda.synchronize()
h = haar.cpu().numpy().reshape(n * 5, 4096)[:, :7]
names = ["load+row pass", "column pass", "threshold search", "gather", "rank+emit"]
mid = h[h.shape[0] // 4: 3 * h.shape[0] // 4]
for i, nm in enumerate(names):
    v = mid[:, i]
    print(f"{nm:20s} mean {v.mean():8.0f} p10 {np.percentile(v,10):8.0f} p50 {np.percentile(v,50):8.0f} p90 {np.percentile(v,90):8.0f}")
print("total", mid[:, :5].sum(axis=1).mean(), "steps mean", mid[:, 5].mean(), "p90", np.percentile(mid[:, 5], 90), "candidates mean", mid[:, 6].mean())
