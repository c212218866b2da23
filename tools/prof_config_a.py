import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
import lbaudiodetective_amd as lb
det = lb.Detective()
clips = lb.synth_clips_device(0x4C424144, 0, 4000, 5512, 5512*9)
for _ in range(2):
    out = det.fingerprint_clips_device(clips)
torch.cuda.synchronize()
