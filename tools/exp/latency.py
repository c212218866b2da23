"""Host-to-host latency of the drop-in entry points on one clip (ProcessPCM, ComparePCM, Stream)."""
import time
import numpy as np
import lbaudiodetective_amd as lb
from oracle import oracle as O

det = lb.Detective().configure(sample_rate=44100.0, window=1024, stride=64)
pcm = O.synth_clip(0x4C424144, 1, 44100, 44100)
pcm9 = np.concatenate([O.synth_clip(0x4C424144, i, 44100, 44100) for i in range(9)])
for name, x in (("1 s", pcm), ("9 s", pcm9)):
    for _ in range(5):
        fp = det.process_pcm(x)
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        fp = det.process_pcm(x)
    dt = (time.perf_counter() - t0) / n
    print(f"process_pcm {name}: {dt * 1e6:.1f} us, {fp.number_of_subfingerprints} sub-fingerprints")
t0 = time.perf_counter()
for _ in range(100):
    m = det.compare_pcm(pcm9, pcm9[44100 * 2: 44100 * 6])
print(f"compare_pcm 9 s vs 4 s: {(time.perf_counter() - t0) / 100 * 1e6:.1f} us, match {m:.3f}")
