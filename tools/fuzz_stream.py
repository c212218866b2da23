#!/usr/bin/env python3
"""Randomized sweep of the host-buffer entry points against the oracle (not part of the test suite):
tools/fuzz_stream.py [trials] [seed].  Streaming (random chunkings, empty and tiny chunks included) must equal
ProcessPCM on the concatenation and the oracle; the host batch call must give the oracle's bits for float32, int16
and int32 clips.
Round 2: 60 000 trials (seed 3), 0 mismatches, 452 s on one MI355X."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lbaudiodetective_amd as lb
from oracle import oracle as O

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
t0 = time.time()
for t in range(trials):
    kind = rng.integers(0, 4)
    if kind == 0:
        cfg = O.Config(44100, 1024)
    elif kind == 1:
        cfg = O.Config(5512, 2048)
    elif kind == 2:
        cfg = O.Config(48000, 4096)
    else:
        cfg = O.Config(float(rng.choice([8000, 11025, 16000, 22050])), int(2 ** rng.integers(5, 11)), int(rng.integers(8, 200)),
                       int(rng.integers(1, 65)), 1)
        cfg.subfp_len = int(rng.integers(1, min(256, 128 * cfg.bands) + 1))
    det = lb.Detective().configure(sample_rate=cfg.sample_rate, window=cfg.window, stride=cfg.stride, bands=cfg.bands,
                                   subfp_len=cfg.subfp_len)
    total = int(rng.integers(0, cfg.window + cfg.stride * 128 * 3 + 1000))
    pcm = O.synth_clip(int(rng.integers(0, 2**31)), 7, 44100, max(total, 1))[:total]
    if rng.integers(0, 2) == 0:
        # ---- streaming ----
        st = lb.Stream(det)
        at, emitted = 0, 0
        while at < total:
            c = int(rng.choice([0, 1, int(rng.integers(1, 100)), int(rng.integers(100, 5000)), int(rng.integers(5000, 60000))]))
            c = min(c, total - at)
            emitted += st.push(pcm[at:at + c])
            at += c
            if emitted != O.subfingerprint_count(at, cfg.window, cfg.stride):
                bad += 1
                print("STREAM COUNT MISMATCH", t, cfg.sample_rate, cfg.window, cfg.stride, cfg.bands, total, at, flush=True)
                break
        want = O.fingerprint_pcm(pcm, cfg) if total >= cfg.window else np.zeros((0, cfg.subfp_len), np.uint8)
        got = st.fingerprint().to_bools() if emitted else np.zeros((0, cfg.subfp_len), np.uint8)
        whole = det.process_pcm(pcm).to_bools() if want.shape[0] else got
        if got.shape[0] != want.shape[0] or (want.shape[0] and not (np.array_equal(got, want) and np.array_equal(whole, want))):
            bad += 1
            print("STREAM MISMATCH", t, cfg.sample_rate, cfg.window, cfg.stride, cfg.bands, cfg.subfp_len, total, flush=True)
        continue
    # ---- host batch, three sample formats ----
    n_clips = int(rng.integers(1, 6))
    spc = max(total, cfg.window + cfg.stride * 128)
    clips = O.synth_clips(int(rng.integers(0, 2**31)), 0, n_clips, 44100, spc)       # multiples of 1 / 32768
    want = O.fingerprint_batch(clips, cfg)
    got = det.fingerprint_clips(clips)
    i16 = np.round(clips * 32768).astype(np.int16)
    got16 = det.fingerprint_clips(i16)
    raw32 = rng.integers(-2**31, 2**31 - 1, clips.shape, dtype=np.int64).astype(np.int32)
    want32 = O.fingerprint_batch((raw32.astype(np.float64) / 2**31).astype(np.float32), cfg)
    got32 = det.fingerprint_clips(raw32)
    if not (np.array_equal(got, want) and np.array_equal(got16, want) and np.array_equal(got32, want32)):
        bad += 1
        print("HOST BATCH MISMATCH", t, cfg.sample_rate, cfg.window, cfg.stride, cfg.bands, cfg.subfp_len, n_clips, spc, flush=True)
print(f"{trials} trials, {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
