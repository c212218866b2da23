"""CompareAudioURLs / ProcessAudioURL latency with the library given on the command line (A/B of experiment builds)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lbaudiodetective_amd import _native as N
if len(sys.argv) > 1:
    N.LIB_PATH = os.path.abspath(sys.argv[1])
import lbaudiodetective_amd as lb
birds = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "birds")
a, b = os.path.join(birds, "BlackBird.caf"), os.path.join(birds, "BlackBird_eql.caf")
det = lb.Detective()
for _ in range(5):
    det.compare_audio_urls(a, b)
for rnd in range(3):
    t = time.perf_counter()
    for _ in range(200):
        det.compare_audio_urls(a, b)
    c = (time.perf_counter() - t) * 1e3 / 200
    t = time.perf_counter()
    for _ in range(200):
        det.process_audio_url(a)
    p = (time.perf_counter() - t) * 1e3 / 200
    print(f"{os.path.basename(N.LIB_PATH)}: CompareAudioURLs {c:.3f} ms, ProcessAudioURL (9 s file) {p:.3f} ms")
