#!/usr/bin/env python3
"""Long randomized GPU-vs-oracle sweep (not part of the test suite): tools/fuzz_parity.py [trials] [seed].
Round 2, final kernels: 40 000 trials (seed 2026) and 6 000 (seed 512) without a mismatch; 150 000 (seed 777) found three
file-loop cases, one bug (tail mode 1 with a zero band divisor); after the fix 150 000 more (seed 31337) and, on the
round's final build, another 150 000 (seed 424242) without a mismatch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lbaudiodetective_amd as lb
from oracle import oracle as O

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
t0 = time.time()
for t in range(trials):
    kind = rng.integers(0, 9)
    if kind == 1:      # stride 64 with 256- .. 2048-sample windows: k_rows_full.hip, or k_rows_stream2.hip (2048, even length, <= 32 bands)
        # (round 3: any even stride on k_rows_full.hip -- 8 is the file hop of 44.1 kHz material at the defaults)
        stride = int(rng.choice([64, 64, 64, 2, 6, 8, 8, 16, 32, 44, 100, 128, 200, 254]))
        cfg = O.Config(float(rng.choice([5512, 8000, 11025, 16000, 22050, 32000, 44100, 48000])), int(rng.choice([256, 512, 1024, 2048])),
                       stride, int(rng.integers(1, 65)), 1)
        cfg.subfp_len = int(rng.integers(1, min(256, 128 * cfg.bands) + 1))
        n = cfg.window + stride * 128 * int(rng.integers(1, 3)) + int(rng.integers(0, 128 * stride))
        clips = int(rng.integers(1, 40))                                                 # (2048: k_rows_stream2.hip walks pairs of runs)
    elif kind in (6, 7):   # 4096-sample windows at stride 64: the streaming kernel (k_rows_stream.hip) or its fallbacks
        cfg = O.Config(float(rng.choice([48000, 44100, 32000, 96000, 22050])), 4096, 64, int(rng.choice([32, 32, 32, 16, 31, 33, 64])), 1)
        cfg.subfp_len = int(rng.integers(1, min(256, 128 * cfg.bands) + 1))
        n = 4096 + 64 * 128 * int(rng.integers(1, 4)) + int(rng.integers(0, 8192))      # even and odd lengths
        clips = int(rng.integers(1, 40))
    elif kind == 8:      # upstream's file loop with a random end-of-file treatment
        cfg = O.Config(float(rng.choice([5512, 8000, 11025, 44100])), int(2 ** rng.integers(6, 12)), int(rng.choice([16, 64, 100])),
                       int(rng.integers(1, 65)), 1)
        cfg.subfp_len = int(rng.integers(1, min(256, 128 * cfg.bands) + 1))
        hop = int(rng.integers(1, 70))
        n_client = int(rng.integers(1, 40000))
        file_frames = int(rng.integers(cfg.window, cfg.window + cfg.stride * 128 * 4))
        tail_mode = int(rng.integers(0, 3))
        x = O.synth_clip(int(rng.integers(0, 2**31)), 5, 44100, n_client)
        det = lb.Detective().configure(sample_rate=cfg.sample_rate, window=cfg.window, stride=cfg.stride, bands=cfg.bands,
                                       subfp_len=cfg.subfp_len)
        det.set_file_tail_mode(tail_mode)
        got = det.process_file_stream(x, file_frames, hop).to_bools()
        want = O.fingerprint_file_loop(x, file_frames, hop, cfg, tail_mode)
        # (no frame at all: an empty fingerprint has no length yet, D.m:297)
        if got.shape[0] != want.shape[0] or (want.shape[0] and not np.array_equal(got, want)):
            bad += 1
            print("FILE LOOP MISMATCH", t, cfg.sample_rate, cfg.window, cfg.stride, cfg.bands, cfg.subfp_len, hop, n_client, file_frames, tail_mode, flush=True)
        continue
    elif kind == 0:      # config B shape through the specialised kernels, odd clip counts / lengths / content
        cfg = O.Config(44100, 1024)
        n = 1024 + 64 * 128 * int(rng.integers(1, 4)) + int(rng.integers(0, 8192))
        clips = int(rng.integers(1, 9))
    else:
        cfg = O.Config(float(rng.choice([5512, 8000, 16000, 22050, 44100, 48000])), int(2 ** rng.integers(4, 13)),
                       int(rng.integers(1, 300)), int(rng.integers(1, 65)), 1)
        cfg.subfp_len = int(rng.integers(1, min(256, 128 * cfg.bands) + 1))
        n = cfg.window + cfg.stride * 128 * int(rng.integers(1, 3)) + int(rng.integers(0, 128 * cfg.stride))
        clips = int(rng.integers(1, 4))
    pcm = O.synth_clips(int(rng.integers(0, 2**31)), int(rng.integers(0, 1000)), clips, 44100, n)
    mode = rng.integers(0, 6)
    if mode == 0: pcm *= np.float32(1e-3)
    if mode == 1: pcm[:, ::2] = 0
    if mode == 2: pcm[:, : n // 3] = pcm[:, n // 3: 2 * (n // 3)]      # repeated content: many equal coefficients
    if mode == 3: pcm = np.sign(pcm).astype(np.float32)
    want = O.fingerprint_batch(pcm, cfg, nthreads=8)
    only = os.environ.get("FUZZ_ONLY")
    if only is not None and int(only) != t:           # replay: same random stream, GPU work for one trial only
        if want.shape[1] >= 1 and clips >= 2:
            rng.integers(1, cfg.subfp_len + 3)
        continue
    if only is not None:                               # ... and a stage-by-stage diagnosis of that trial
        det = lb.Detective().configure(sample_rate=cfg.sample_rate, window=cfg.window, stride=cfg.stride,
                                       bands=cfg.bands, subfp_len=cfg.subfp_len)
        out, raw, haar = det.fingerprint_clips_device(torch.from_numpy(pcm).cuda(), taps=True)
        raw, haar = raw.cpu().numpy(), haar.cpu().numpy()
        for c in range(clips):
            obits, oraw, ohaar = O.fingerprint_pcm(pcm[c], cfg, taps=True)
            dr = np.argwhere(raw[c].view(np.uint32) != oraw.view(np.uint32))
            dh = np.argwhere(haar[c].view(np.uint32) != ohaar.view(np.uint32))
            print("clip", c, "raw diffs", len(dr), dr[:5].tolist(), "haar diffs", len(dh), dh[:5].tolist())
            for f, rw, b in dh[:6]:
                print("   haar", (f, rw, b), haar[c][f, rw, b], ohaar[f, rw, b], hex(haar[c][f, rw, b].view(np.uint32)), hex(ohaar[f, rw, b].view(np.uint32)), "raw row", oraw[f, rw].tolist()[:14])
            print("   bits differ at", np.argwhere(lb.unpack_packed(out[c].cpu().numpy(), cfg.subfp_len) != obits)[:10].tolist())
            for f, rw, b in dr[:5]:
                print("   raw", (f, rw, b), raw[c][f, rw, b], oraw[f, rw, b], hex(raw[c][f, rw, b].view(np.uint32)), hex(oraw[f, rw, b].view(np.uint32)))
    det = lb.Detective().configure(sample_rate=cfg.sample_rate, window=cfg.window, stride=cfg.stride, bands=cfg.bands,
                                   subfp_len=cfg.subfp_len)
    packed = det.fingerprint_clips_device(torch.from_numpy(pcm).cuda())
    got = lb.unpack_packed(packed.cpu().numpy(), cfg.subfp_len).reshape(want.shape)
    if not np.array_equal(got, want):
        bad += 1
        print("MISMATCH", t, kind, mode, cfg.sample_rate, cfg.window, cfg.stride, cfg.bands, cfg.subfp_len, n, clips, flush=True)
    # compare leg on the fresh fingerprints
    if want.shape[1] >= 1 and clips >= 2:
        a, b = want[0], want[1]
        rg = int(rng.integers(1, cfg.subfp_len + 3))
        g = np.float32(lb.Fingerprint.from_bools(a).compare_to_fingerprint(lb.Fingerprint.from_bools(b), rg))
        w = np.float32(O.compare_fp(a, b, rg))
        if g.view(np.uint32) != w.view(np.uint32):
            bad += 1
            print("COMPARE MISMATCH", t, rg, g, w, flush=True)
print(f"{trials} trials, {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
