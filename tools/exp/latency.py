"""Host-to-host latency of the drop-in entry points on one clip (ProcessPCM, ComparePCM, Stream, one-off
fingerprint compare).  Uses only calls that exist since round 1, so it can be pointed at an older tree:
    PYTHONPATH=<tree> python tools/exp/latency.py"""
import json
import time
import numpy as np
import lbaudiodetective_amd as lb
from oracle import oracle as O

det = lb.Detective().configure(sample_rate=44100.0, window=1024, stride=64)
pcm = O.synth_clip(0x4C424144, 1, 44100, 44100)
pcm9 = np.concatenate([O.synth_clip(0x4C424144, i, 44100, 44100) for i in range(9)])
out = {"library": lb.LIB_PATH}
for name, x in (("1s", pcm), ("9s", pcm9)):
    for _ in range(5):
        fp = det.process_pcm(x)
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        fp = det.process_pcm(x)
    out[f"process_pcm_{name}_us"] = round((time.perf_counter() - t0) / n * 1e6, 1)
t0 = time.perf_counter()
for _ in range(100):
    m = det.compare_pcm(pcm9, pcm9[44100 * 2: 44100 * 6])
out["compare_pcm_9s_vs_4s_us"] = round((time.perf_counter() - t0) / 100 * 1e6, 1)
# the birds' shape: 48 against 21 sub-fingerprints, 28 sliding offsets
rng = np.random.default_rng(3)
def rand_fp(n):
    pos = rng.random((n, 100)) < 0.5
    f = np.zeros((n, 200), np.uint8)
    f[:, 0::2], f[:, 1::2] = pos, ~pos
    return lb.Fingerprint.from_bools(f)
a, b = rand_fp(48), rand_fp(21)
for _ in range(5):
    a.compare_to_fingerprint(b, 200)
t0 = time.perf_counter()
for _ in range(200):
    a.compare_to_fingerprint(b, 200)
out["compare_to_fingerprint_48x21_us"] = round((time.perf_counter() - t0) / 200 * 1e6, 1)
a, b = rand_fp(1500), rand_fp(700)
a.compare_to_fingerprint(b, 200)
t0 = time.perf_counter()
for _ in range(5):
    a.compare_to_fingerprint(b, 200)
out["compare_to_fingerprint_1500x700_us"] = round((time.perf_counter() - t0) / 5 * 1e6, 1)
# streaming: 9 s pushed in 100 ms chunks
st = lb.Stream(det)
t0 = time.perf_counter()
for i in range(0, pcm9.size, 4410):
    st.push(pcm9[i:i + 4410])
out["stream_9s_in_100ms_chunks_us_per_push"] = round((time.perf_counter() - t0) / (pcm9.size // 4410) * 1e6, 1)
print(json.dumps(out))
