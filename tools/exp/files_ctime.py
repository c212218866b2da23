"""The C entry point alone (no Python objects inside the timed region): LBAudioDetectiveProcessAudioURLs on copies x
sixty fixtures; the fingerprints are released outside the clock."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lbaudiodetective_amd import _native as N
if os.environ.get("LBAD_LIB"):
    N.LIB_PATH = os.path.abspath(os.environ["LBAD_LIB"])
import lbaudiodetective_amd as lb
birds = os.path.join(ROOT, "tests", "golden", "birds")
paths = sorted(os.path.join(birds, f) for f in os.listdir(birds) if f.endswith(".caf"))
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
batch = paths * copies
n = len(batch)
det = lb.Detective()
L = det._L
arr = (C.c_char_p * n)(*[p.encode() for p in batch])
for pipe in [int(v) for v in os.environ.get("PIPES", "1,0,1,0").split(",")]:
    L.LBAudioDetectiveSetFilePipeline(det._ref, pipe)
    best, tot = 1e9, 0.0
    for r in range(rounds + 1):
        refs = (N.Ref * n)()
        sts = (N.OSStatus * n)()
        t = time.perf_counter()
        rc = L.LBAudioDetectiveProcessAudioURLs(det._ref, arr, n, refs, sts)
        dt = time.perf_counter() - t
        assert rc == 0
        t = time.perf_counter()
        for i in range(n):
            if refs[i]:
                L.LBAudioDetectiveFingerprintDispose(refs[i])
        rel = time.perf_counter() - t
        if r:
            best, tot = min(best, dt), tot + dt
    print(f"C call, {n} files, pipeline {pipe}: mean {tot / rounds * 1e3:.2f} ms, best {best * 1e3:.2f} ms, {n * rounds / tot:.0f} files/s; "
          f"releasing the fingerprints from Python {rel * 1e3:.2f} ms", flush=True)
