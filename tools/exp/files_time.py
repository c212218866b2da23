"""Files per second of LBAudioDetectiveProcessAudioURLs on the sixty bird fixtures (x copies) and of the two-file
CompareAudioURLs; run under rocprofv3 --kernel-trace --stats for the device share."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lbaudiodetective_amd as lb
birds = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "birds")
paths = sorted(os.path.join(birds, f) for f in os.listdir(birds) if f.endswith(".caf"))
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
batch = paths * copies
det = lb.Detective()
det.process_audio_urls(batch)
t = time.perf_counter()
for _ in range(rounds):
    fps = det.process_audio_urls(batch)
dt = time.perf_counter() - t
print(f"{len(batch)} files per call: {len(batch) * rounds / dt:.0f} files/s, {dt * 1e3 / rounds:.2f} ms per call, {dt * 1e6 / rounds / len(batch):.1f} us per file")
a, b = os.path.join(birds, "BlackBird.caf"), os.path.join(birds, "BlackBird_eql.caf")
det.compare_audio_urls(a, b)
t = time.perf_counter()
for _ in range(50):
    det.compare_audio_urls(a, b)
print(f"CompareAudioURLs: {(time.perf_counter() - t) * 1e3 / 50:.3f} ms")
t = time.perf_counter()
for _ in range(rounds):
    for p in batch:
        open(p, "rb").read()
print(f"reading the files alone: {(time.perf_counter() - t) * 1e3 / rounds:.2f} ms per {len(batch)} files")
