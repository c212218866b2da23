import os, sys
sys.path.insert(0, "/root/repo")
from lbaudiodetective_amd import _native as N
N.LIB_PATH = "/root/repo/lbaudiodetective_amd/lib/exp/lib_ft.so"
import lbaudiodetective_amd as lb
birds = "/root/repo/tests/golden/birds"
paths = sorted(os.path.join(birds, f) for f in os.listdir(birds) if f.endswith(".caf"))
det = lb.Detective()
for _ in range(4):
    det.process_audio_urls(paths)
