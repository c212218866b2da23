#!/usr/bin/env python3
"""Driver for rocprofv3 counter passes: a few launches of the fingerprint path on N clips, plus a
device-to-device copy of known size (calibrates FETCH_SIZE / WRITE_SIZE, see MI355X_MICROARCH.md HBM)
and one corpus scan."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lbaudiodetective_amd as lb

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n_corpus = int(sys.argv[4]) if len(sys.argv) > 4 else 0
det = lb.Detective().configure(sample_rate=44100, window=1024)
det.set_kernel_variant(variant)
clips = lb.synth_clips_device(0x4C424144, 0, n, 44100, 44100)
out = None
for _ in range(reps):
    out = det.fingerprint_clips_device(clips, out=out)
torch.cuda.synchronize()
# calibration copy: reads and writes exactly clips.numel() * 4 bytes
dst = torch.empty_like(clips)
for _ in range(2):
    dst.copy_(clips)
torch.cuda.synchronize()
if n_corpus:
    corpus = lb.Corpus(200, 5, n_corpus)
    step = 1 << 20
    for b in range(0, n_corpus, step):
        corpus.append_packed_device(lb.synth_corpus_device(0x4C424145, b, min(step, n_corpus - b), 5, 200))
    q = lb.Fingerprint.from_bools(lb.unpack_packed(lb.synth_corpus_device(0x4C424145, 777, 1, 5, 200).cpu().numpy(), 200))
    for _ in range(5):
        r = corpus.query(q)
    print("query", r)
print("done", n, variant, clips.numel() * 4)
