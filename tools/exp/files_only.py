"""Only the batch call on the sixty fixtures (for rocprofv3 --kernel-trace --stats: every kernel launch is one of a batch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lbaudiodetective_amd as lb
birds = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "birds")
paths = sorted(os.path.join(birds, f) for f in os.listdir(birds) if f.endswith(".caf"))
det = lb.Detective()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    det.process_audio_urls(paths)
