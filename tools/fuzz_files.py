#!/usr/bin/env python3
"""Randomized sweep of the file entry points (not part of the test suite): tools/fuzz_files.py [trials] [seed].
Random WAV and CAF files (rate, 1-3 channels, 8 / 16 / 24 / 32-bit integer or 32 / 64-bit float samples of either
byte order, IMA4 packets with and without a packet table, any length) through LBAudioDetectiveProcessAudioURL --
decode and conversion to the processing rate on the DEVICE, then upstream's file loop -- against the ORACLE's own
file front end (oracle/lbad_file_oracle.c: container, IMA4 / LPCM decode, converter models, file loop; no code shared
with the library): the two must agree bit for bit for every payload format, rate ratio (decimating and
interpolating), converter model, hop mode and end-of-file treatment.  The library's HOST functions
(LBAudioDetectiveReadAudioURL) are checked against the oracle's decode + conversion on the way.
Round 2, decode and conversion on the device: 60 000 trials (seed 13), 0 mismatches, 423 s on one MI355X (34 570 of the
files long enough for at least one sub-fingerprint: 9 793 IMA4, 12 483 CAF LPCM, 12 294 WAV)."""
import os, struct, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lbaudiodetective_amd as lb
from oracle import oracle as O

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
rejected = {}
nonempty = {}
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


def write_caf(rate, fourcc, flags, bytes_per_packet, frames_per_packet, channels, bits, payload, pakt=None):
    desc = struct.pack(">d4sIIIII", float(rate), fourcc, flags, bytes_per_packet, frames_per_packet, channels, bits)
    out = b"caff" + struct.pack(">HH", 1, 0) + b"desc" + struct.pack(">q", len(desc)) + desc
    if pakt is not None:
        body = struct.pack(">qqii", pakt[0], pakt[1], pakt[2], 0)
        out += b"pakt" + struct.pack(">q", len(body)) + body
    out += b"data" + struct.pack(">q", 4 + len(payload)) + struct.pack(">I", 0) + payload
    open(path, "wb").write(out)


def write_caf_lpcm(x, rate, channels, kind, little):
    """x: float64 [frames, channels]; kind: i8 / i16 / i24 / i32 / f32 / f64."""
    e = "<" if little else ">"
    if kind == "f32":
        data, bits, fl = x.astype(e + "f4").tobytes(), 32, 1
    elif kind == "f64":
        data, bits, fl = x.astype(e + "f8").tobytes(), 64, 1
    elif kind == "i8":
        data, bits, fl = np.clip(np.round(x * 128), -128, 127).astype(np.int8).tobytes(), 8, 0
    elif kind == "i16":
        data, bits, fl = np.clip(np.round(x * 32768), -32768, 32767).astype(e + "i2").tobytes(), 16, 0
    elif kind == "i24":
        v = np.clip(np.round(x * 8388608), -8388608, 8388607).astype("<i4").view(np.uint8).reshape(-1, 4)[:, :3]
        data, bits, fl = (v if little else v[:, ::-1]).tobytes(), 24, 0
    else:
        data, bits, fl = np.clip(np.round(x * 2147483648.0), -2**31, 2**31 - 1).astype(e + "i4").tobytes(), 32, 0
    write_caf(rate, b"lpcm", fl | (2 if little else 0), channels * bits // 8, 1, channels, bits, data)


BIRD = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "birds", "BlackBird.caf"), "rb").read()
_at = BIRD.index(b"data") + 12 + 4
IMA_PACKETS = np.frombuffer(BIRD[_at: _at + 34 * ((len(BIRD) - _at) // 34)], np.uint8).reshape(-1, 34)


def write_caf_ima4(rate, channels, n_packets):
    """Any 34-byte packet is a valid IMA4 packet: interleave packets of a bird fixture as `channels` channels."""
    pick = rng.integers(0, IMA_PACKETS.shape[0], size=n_packets * channels)
    payload = IMA_PACKETS[pick].tobytes()
    pakt = None
    if rng.integers(0, 2):
        priming = int(rng.choice([0, 0, 7, 64, 100]))
        valid = int(rng.integers(0, n_packets * 64 + 1))
        pakt = (n_packets, valid, priming)
    write_caf(rate, b"ima4", 0, 34 * channels, 64, channels, 0, payload, pakt)


for t in range(trials):
    file_rate = int(rng.choice([4000, 8000, 11025, 16000, 22050, 32000, 44100, 48000, 96000]))
    channels = int(rng.choice([1, 1, 2, 3]))
    container = rng.integers(0, 3)
    frames = int(rng.integers(0, 3 * file_rate))
    if container == 2:
        write_caf_ima4(file_rate, channels, frames // 64)
    else:
        x = O.synth_clip(int(rng.integers(0, 2**31)), 3, 44100, max(frames * channels, 1))[: frames * channels].astype(np.float64).reshape(frames, channels)
        if container == 0:
            write_wav(x, file_rate, channels, str(rng.choice(["f32", "u8", "i16", "i24", "i32"])))
        else:
            write_caf_lpcm(x, file_rate, channels, str(rng.choice(["f32", "f64", "i8", "i16", "i24", "i32"])), bool(rng.integers(0, 2)))
    kind = ("wav", "caf-lpcm", "caf-ima4")[container]
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
        want = O.fingerprint_file(path, cfg, hop_mode, tail_mode, resampler)
        xs, rate = O.decode_audio_file(path)
        hs, hrate = lb.read_audio_url(path)                      # the library's host decoder and converter, on the way
        hy, _ = lb.read_audio_url(path, cfg.sample_rate, resampler)
        if hrate != rate or not np.array_equal(hs, xs) or not np.array_equal(hy, O.resample(xs, rate, cfg.sample_rate, resampler)):
            bad += 1
            print("HOST FRONT END MISMATCH", t, file_rate, channels, kind, frames, cfg.sample_rate, resampler, flush=True)
    except (ValueError, FileNotFoundError):
        want = ("error",)
    if isinstance(got, tuple):
        got = ("error",)
    same = (isinstance(got, tuple) and isinstance(want, tuple) and got == want) or \
           (not isinstance(got, tuple) and not isinstance(want, tuple) and got.shape[0] == want.shape[0] and (want.shape[0] == 0 or np.array_equal(got, want)))
    if isinstance(got, tuple):
        rejected[kind] = rejected.get(kind, 0) + 1
    elif got.shape[0]:
        nonempty[kind] = nonempty.get(kind, 0) + 1
    if not same:
        bad += 1
        print("FILE MISMATCH", t, file_rate, channels, kind, frames, cfg.sample_rate, cfg.window, cfg.stride, cfg.bands, cfg.subfp_len,
              hop_mode, tail_mode, resampler, got if isinstance(got, tuple) else got.shape, want if isinstance(want, tuple) else want.shape, flush=True)
print(f"{trials} trials, {bad} mismatches, {time.time() - t0:.1f} s; files with at least one sub-fingerprint {nonempty}, rejected {rejected}")
sys.exit(1 if bad else 0)
