import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
import lbaudiodetective_amd as lb
from oracle import oracle as O
for rate in (48000, 16000, 8000):
    clips = lb.synth_clips_device(0x4C424144, 3, 4, rate, 1024 + 64*128*2)
    want = O.fingerprint_batch(clips.cpu().numpy(), O.Config(rate, 1024), nthreads=4)
    idx, lo, hi = O.band_table(rate, 1024)
    print(rate, "live", [b for b in range(32) if lo[b] < hi[b]], "layout", None)
    for v in (0, 4, 1):
        det = lb.Detective().configure(sample_rate=rate, window=1024)
        print("  layout", lb.compact_layout(det))
        det.set_kernel_variant(v)
        got = lb.unpack_packed(det.fingerprint_clips_device(clips).cpu().numpy(), 200).reshape(want.shape)
        print("  variant", v, "equal oracle", np.array_equal(got, want), "bits set", int(got.sum()), int(want.sum()))
