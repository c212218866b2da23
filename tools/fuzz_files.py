#!/usr/bin/env python3
"""Randomized sweep of the file entry points (not part of the test suite): tools/fuzz_files.py [trials] [seed].
Random WAV files (rate, channels, 8 / 16 / 24 / 32-bit integer or float samples, length) through
LBAudioDetectiveProcessAudioURL -- decode on the host, conversion to the processing rate on the DEVICE, upstream's
file loop -- against the oracle fed by the library's HOST decoder + converter (LBAudioDetectiveReadAudioURL): the two
converters must agree bit for bit for every rate ratio (decimating and interpolating), converter model, hop mode and
end-of-file treatment.
Round 2: 60 000 trials (seed 7), 0 mismatches, 489 s on one MI355X."""
import os, struct, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lbaudiodetective_amd as lb
from oracle import oracle as O

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
t0 = time.time()
tmp = tempfile.mkdtemp()
path = os.path.join(tmp, "f.wav")


def write_wav(x, rate, channels, kind):
    """x: float64 [frames, channels] in [-1, 1)."""
    if kind == "f32":
        data, tag, bits = x.astype("<f4").tobytes(), 3, 32
    elif kind == "u8":
        data, tag, bits = (np.clip(np.round(x * 128) + 128, 0, 255)).astype(np.uint8).tobytes(), 1, 8
    elif kind == "i16":
        data, tag, bits = np.clip(np.round(x * 32768), -32768, 32767).astype("<i2").tobytes(), 1, 16
    elif kind == "i24":
        v = np.clip(np.round(x * 8388608), -8388608, 8388607).astype("<i4")
        data, tag, bits = v.view(np.uint8).reshape(-1, 4)[:, :3].tobytes(), 1, 24
    else:
        data, tag, bits = np.clip(np.round(x * 2147483648.0), -2**31, 2**31 - 1).astype("<i4").tobytes(), 1, 32
    block = channels * bits // 8
    hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, tag, channels, rate, rate * block, block, bits)
    open(path, "wb").write(hdr + b"data" + struct.pack("<I", len(data)) + data)


for t in range(trials):
    file_rate = int(rng.choice([4000, 8000, 11025, 16000, 22050, 32000, 44100, 48000, 96000]))
    channels = int(rng.choice([1, 1, 2, 3]))
    kind = str(rng.choice(["f32", "u8", "i16", "i24", "i32"]))
    frames = int(rng.integers(0, 3 * file_rate))
    x = O.synth_clip(int(rng.integers(0, 2**31)), 3, 44100, max(frames * channels, 1))[: frames * channels].astype(np.float64).reshape(frames, channels)
    write_wav(x, file_rate, channels, kind)
    if rng.integers(0, 3) == 0:
        cfg = O.Config(float(rng.choice([5512, 8000, 11025, 44100])), int(2 ** rng.integers(7, 12)), int(rng.choice([32, 64, 100])),
                       int(rng.integers(1, 65)), 1)
        cfg.subfp_len = int(rng.integers(1, min(256, 128 * cfg.bands) + 1))
    else:
        cfg = O.Config()
    hop_mode, tail_mode, resampler = int(rng.integers(0, 2)), int(rng.integers(0, 3)), int(rng.integers(0, 3))
    det = lb.Detective().configure(sample_rate=cfg.sample_rate, window=cfg.window, stride=cfg.stride, bands=cfg.bands,
                                   subfp_len=cfg.subfp_len)
    det.set_file_hop_mode(hop_mode).set_file_tail_mode(tail_mode)
    det.set_resampler_mode(resampler)
    try:
        got = det.process_audio_url(path).to_bools()
    except lb.LBAudioDetectiveError as e:
        got = ("error", e.status)
    try:
        xs, rate = lb.read_audio_url(path)
        y, _ = lb.read_audio_url(path, cfg.sample_rate, resampler)
        if hop_mode == 0:
            want = O.fingerprint_pcm(y, cfg) if y.size >= cfg.window else np.zeros((0, cfg.subfp_len), np.uint8)
        else:
            hop = max(1, int(round(cfg.stride * cfg.sample_rate / rate)))
            want = O.fingerprint_file_loop(y, xs.size, hop, cfg, tail_mode)
    except lb.LBAudioDetectiveError as e:
        want = ("error", e.status)
    same = (isinstance(got, tuple) and isinstance(want, tuple) and got == want) or \
           (not isinstance(got, tuple) and not isinstance(want, tuple) and got.shape[0] == want.shape[0] and (want.shape[0] == 0 or np.array_equal(got, want)))
    if not same:
        bad += 1
        print("FILE MISMATCH", t, file_rate, channels, kind, frames, cfg.sample_rate, cfg.window, cfg.stride, cfg.bands, cfg.subfp_len,
              hop_mode, tail_mode, resampler, got if isinstance(got, tuple) else got.shape, want if isinstance(want, tuple) else want.shape, flush=True)
print(f"{trials} trials, {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
