"""Host-to-host time of the file entry points on two bundled bird files (BASELINE configs[0]):
    python tools/exp/file_latency.py
decode (CAF / IMA4), resample 44.1 kHz -> 5512 Hz (each of the three converter models), fingerprint, compare."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lbaudiodetective_amd as lb
B = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "birds")
a, b = os.path.join(B, "BlackBird.caf"), os.path.join(B, "BlackBird_eql.caf")


def best(f, n=5):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t)
    return min(ts) * 1e3, r


print("cores", os.cpu_count())
print("decode only            %.2f ms" % best(lambda: lb.read_audio_url(a, 0, 0))[0])
for mode, name in ((0, "long sinc"), (1, "short sinc"), (2, "linear")):
    print("decode + resample %-10s %.2f ms" % (name, best(lambda: lb.read_audio_url(a, 5512, mode))[0]))
d = lb.Detective()
d.compare_audio_urls(a, b)
for mode in (0, 1, 2):
    d.set_resampler_mode(mode)
    ms, m = best(lambda: d.compare_audio_urls(a, b))
    print("CompareAudioURLs (9 s vs 4 s), resampler %d: %.2f ms, match %.4f" % (mode, ms, m))
x, _ = lb.read_audio_url(a, 5512, 0)
y, _ = lb.read_audio_url(b, 5512, 0)
print("ComparePCM on the converted samples: %.3f ms" % best(lambda: d.compare_pcm(x, y), 20)[0])
