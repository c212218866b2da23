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
ref = None
for rep in range(2):                                         # interleaved A/B: pipeline on, off, on, off
    for pipe in (True, False):
        det.set_file_pipeline(pipe)
        fps = det.process_audio_urls(batch)
        bits = [f.to_bools().tobytes() for f in fps[:60]]
        ref = ref or bits
        assert bits == ref, "results depend on the pipeline"
        t = time.perf_counter()
        for _ in range(rounds):
            fps = det.process_audio_urls(batch)
        dt = time.perf_counter() - t
        print(f"{len(batch)} files per call, pipeline {'on ' if pipe else 'off'}: {len(batch) * rounds / dt:.0f} files/s, "
              f"{dt * 1e3 / rounds:.2f} ms per call, {dt * 1e6 / rounds / len(batch):.1f} us per file")
det.set_file_pipeline(True)
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
