"""Quick check + timing of the streaming stage-1 kernels: python tools/exp/stream_check.py [C|A] [--time]
C = 48 kHz / 4096 (configs[4]), A = 5512 Hz / 2048 (the defaults)."""
import sys, numpy as np, torch
import lbaudiodetective_amd as lb
from oracle import oracle as O
SEED = 0x4C424144
which = "A" if "A" in sys.argv[1:] else "C"
rate, window, stereo, secs, nbig = (48000, 4096, True, 1, 10000) if which == "C" else (5512, 2048, False, 9, 20000)
cfg = O.Config(rate, window)
n = cfg.window + 64 * (128 * 2 + 17) + 1
n -= n & 1
pcm = O.synth_clips(SEED, 100, 3, rate, n, stereo)
det = lb.Detective().configure(sample_rate=rate, window=window)
ok = True
for variant in ((2, 3, 1) if which == "A" else (2, 1)):
    det.set_kernel_variant(variant)
    clips = torch.from_numpy(pcm).cuda()
    bits, raw, haar = det.fingerprint_clips_device(clips, taps=True)
    torch.cuda.synchronize()
    raw = raw.cpu().numpy()
    for c in range(3):
        obits, oraw, ohaar = O.fingerprint_pcm(pcm[c], cfg, taps=True)
        same = np.array_equal(raw[c], oraw)
        if not same:
            ok = False
            d = raw[c] != oraw
            print(f"variant {variant} clip {c}: {d.sum()} of {d.size} row values differ; frames {np.unique(np.nonzero(d)[0])}, "
                  f"rows {np.unique(np.nonzero(d)[1])[:20]}, bands {np.unique(np.nonzero(d)[2])}")
            i = tuple(np.argwhere(d)[0])
            print("   first", i, raw[c][i], oraw[i])
        else:
            print(f"variant {variant} clip {c}: rows bit-exact")
if "--time" in sys.argv:
    nclips = nbig
    big = lb.synth_clips_device(SEED, 0, nclips, rate, rate * secs, stereo)
    for variant in ((2, 3, 1) if which == "A" else (2, 1)):
        det.set_kernel_variant(variant)
        out = det.fingerprint_clips_device(big)
        torch.cuda.synchronize()
        det.set_stage_timing(True)
        for _ in range(5):
            det.fingerprint_clips_device(big, out=out)
        s1, s2, launches = det.stage_times()
        det.set_stage_timing(False)
        print(f"variant {variant}: stage 1 {s1 / launches:.3f} ms, stage 2 {s2 / launches:.3f} ms per {nclips} clips")
sys.exit(0 if ok else 1)
