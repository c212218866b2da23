"""Device front end (batch kernels, one file) against the oracle, sample by sample, for a synthetic WAV at a given rate."""
import os, struct, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import lbaudiodetective_amd as lb
from oracle import oracle as O
tmp = tempfile.mkdtemp()
for rate, proc, mode, n in [(48000, 5512, 0, 120000), (96000, 5512, 0, 200000), (32000, 5512, 1, 80000), (44100, 5512, 0, 100000),
                            (48000, 11025, 1, 90000), (8000, 5512, 0, 30000), (22050, 5512, 0, 70001)]:
    x = (np.random.default_rng(rate).standard_normal(n) * 6000).astype("<i2")
    p = os.path.join(tmp, "f.wav")
    pcm = x.tobytes()
    open(p, "wb").write(b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, rate, rate * 2, 2, 16)
                        + b"data" + struct.pack("<I", len(pcm)) + pcm)
    det = lb.Detective().configure(sample_rate=proc)
    det.set_resampler_mode(mode)
    got, ff, fr = det.convert_audio_url(p)
    dec, r = O.decode_audio_file(p)
    want = O.resample(dec, r, float(proc), mode)
    bad = np.nonzero(got.view(np.uint32) != want.view(np.uint32))[0] if got.shape == want.shape else None
    print(rate, proc, mode, got.shape, want.shape, "mismatches", None if bad is None else (bad.size, bad[:8], got[bad[:4]], want[bad[:4]]))
