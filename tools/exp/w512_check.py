import sys, numpy as np, torch
import lbaudiodetective_amd as lb
from oracle import oracle as O
SEED=0x4C424144
for rate, window, nbig in ((22050, 512, 100000), (11025, 256, 100000)):
    cfg = O.Config(rate, window)
    n = window + 64*(128*2+17)
    pcm = O.synth_clips(SEED, 100, 3, rate, n)
    det = lb.Detective().configure(sample_rate=rate, window=window)
    for variant in (0, 1):
        det.set_kernel_variant(variant)
        bits, raw, haar = det.fingerprint_clips_device(torch.from_numpy(pcm).cuda(), taps=True)
        raw = raw.cpu().numpy()
        ok = all(np.array_equal(raw[c], O.fingerprint_pcm(pcm[c], cfg, taps=True)[1], equal_nan=True) for c in range(3))
        big = lb.synth_clips_device(SEED, 0, nbig, rate, rate)
        out = det.fingerprint_clips_device(big); torch.cuda.synchronize()
        det.set_stage_timing(True)
        for _ in range(3): det.fingerprint_clips_device(big, out=out)
        s1, s2, l = det.stage_times(); det.set_stage_timing(False)
        print(rate, window, 'variant', variant, 'rows exact', ok, f'stage1 {s1/l:.3f} ms stage2 {s2/l:.3f} ms')
