"""GPU parity of the file entry points against the oracle's OWN file front end (oracle/lbad_file_oracle.c +
oracle/lbad_oracle.c: container, IMA4 / LPCM decode, converter, upstream's window loop -- no code shared with the
library).  What ExtAudioFile + the loop of LBAudioDetective.m:208-308 do upstream, file by file; here also as one
batch call."""
import os
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BIRDS = os.path.join(os.path.dirname(__file__), "golden", "birds")


def _all_birds():
    return sorted(os.path.join(BIRDS, f) for f in os.listdir(BIRDS) if f.endswith(".caf"))


def test_device_front_end_equals_oracle_on_every_fixture(lb, gpu, oracle):
    """Payload decode and sample-rate conversion ON THE DEVICE (k_decode.hip, k_resample.hip) against the
    independent oracle: all sixty upstream fixtures, every packet, every sample, the three converter models."""
    det = lb.Detective()
    for i, p in enumerate(_all_birds()):
        x, rate = oracle.decode_audio_file(p)
        for mode in ((0, 1, 2) if i % 10 == 0 else (0,)):
            det.set_resampler_mode(mode)
            got, file_frames, file_rate = det.convert_audio_url(p)
            assert file_frames == x.size and file_rate == rate
            assert np.array_equal(got, oracle.resample(x, rate, 5512.0, mode)), (p, mode)
    det.set_resampler_mode(0)
    d2 = lb.Detective().configure(sample_rate=44100)          # equal rates: the decoded samples themselves
    x, _ = oracle.decode_audio_file(_all_birds()[3])
    assert np.array_equal(d2.convert_audio_url(_all_birds()[3])[0], x)
    d3 = lb.Detective().configure(sample_rate=48000)          # interpolating
    assert np.array_equal(d3.convert_audio_url(_all_birds()[3])[0], oracle.resample(x, 44100.0, 48000.0, 0))


def test_ima4_payload_at_an_odd_file_offset_and_stereo(lb, gpu, oracle, tmp_path):
    """The IMA4 decoder reads a packet as 16-bit words when it may (round 4): a `free` chunk of odd length in front of the
    data chunk puts every packet at an odd address (the byte path), and a two-channel file whose packets interleave goes
    through the accumulate-and-average branch.  Decoded samples, converted samples and fingerprints equal the oracle's."""
    import struct
    src = open(os.path.join(BIRDS, "BlackBird.caf"), "rb").read()
    at, chunks = 8, []
    while at + 12 <= len(src):
        size = struct.unpack(">q", src[at + 4:at + 12])[0]
        ln = len(src) - (at + 12) if size < 0 else size
        chunks.append((src[at:at + 4], src[at + 12:at + 12 + ln]))
        at += 12 + ln
    def build(pad, stereo):
        out = src[:8]
        for tag, body in chunks:
            if tag == b"desc" and stereo:                     # same packets, declared as the two channels of a stereo file
                rate, fourcc, flags, bpp, fpp, ch, bits = struct.unpack(">d4sIIIII", body)
                body = struct.pack(">d4sIIIII", rate, fourcc, flags, 2 * bpp, fpp, 2, bits)
            if tag == b"pakt" and stereo:
                n_packets, n_frames, prime, remain = struct.unpack(">qqii", body[:24])
                body = struct.pack(">qqii", n_packets // 2, (n_packets // 2) * 64, 0, 0) + body[24:]
            if tag == b"data":
                out += b"free" + struct.pack(">q", pad) + bytes(pad)
                if stereo:
                    body = body[:4] + body[4:4 + ((len(body) - 4) // 68) * 68]
            out += tag + struct.pack(">q", len(body)) + body
        return out
    det = lb.Detective()
    for name, pad, stereo in (("odd.caf", 7, False), ("even.caf", 8, False), ("odd_stereo.caf", 5, True)):
        path = str(tmp_path / name)
        open(path, "wb").write(build(pad, stereo))
        x, rate = oracle.decode_audio_file(path)
        d44 = lb.Detective().configure(sample_rate=rate)
        assert np.array_equal(d44.convert_audio_url(path)[0], x), name            # the decoded samples themselves
        got, frames, frate = det.convert_audio_url(path)
        assert frames == x.size and frate == rate
        assert np.array_equal(got, oracle.resample(x, rate, 5512.0, 0)), name
        assert np.array_equal(det.process_audio_url(path).to_bools(), oracle.fingerprint_file(path, oracle.Config(), 1, 1, 0)), name


@pytest.mark.parametrize("hop_mode,tail_mode", [(1, 1), (1, 0), (1, 2), (0, 1)])
def test_batch_of_all_fixtures_equals_single_calls_and_oracle(lb, gpu, oracle, hop_mode, tail_mode):
    """LBAudioDetectiveProcessAudioURLs on the sixty fixtures at once == sixty LBAudioDetectiveProcessAudioURL calls
    == the oracle's decode + convert + window loop, bit for bit, for every hop / end-of-file model."""
    paths = _all_birds()
    det = lb.Detective()
    det.set_file_hop_mode(hop_mode).set_file_tail_mode(tail_mode)
    batch = det.process_audio_urls(paths)
    cfg = oracle.Config()
    step = 1 if (hop_mode, tail_mode) == (1, 1) else 4                      # the default model on every file
    for i in range(0, len(paths), step):
        want = oracle.fingerprint_file(paths[i], cfg, hop_mode, tail_mode, 0)
        got = batch[i].to_bools()
        assert got.shape == want.shape and np.array_equal(got, want), (paths[i], hop_mode, tail_mode)
        if i % 12 == 0:
            assert det.process_audio_url(paths[i]).equal_to_fingerprint(batch[i])
    assert sum(f.number_of_subfingerprints for f in batch) == (1518 if hop_mode == 1 else sum(
        oracle.fingerprint_file(p, cfg, 0, 1, 0).shape[0] for p in paths))
    assert det.analysis_stride == 64                                          # the public stride is untouched


def _wav(path, x, rate, channels=1):
    pcm = np.asarray(x, "<i2").tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVE" + b"fmt " +
                struct.pack("<IHHIIHH", 16, 1, channels, rate, rate * 2 * channels, 2 * channels, 16))
        f.write(b"data" + struct.pack("<I", len(pcm)) + pcm)


def test_batch_with_mixed_rates_bad_files_and_other_settings(lb, gpu, oracle, tmp_path):
    """Files of different sample rates (different hops: several launch chains), stereo, too short for one window,
    missing and malformed files in ONE call: per-file statuses, the good ones bit-equal to the oracle; and a
    non-default configuration."""
    rng = np.random.default_rng(31)
    paths, kinds = [], []
    for k, (rate, seconds, ch) in enumerate([(8000, 6, 1), (44100, 3, 2), (22050, 5, 1), (8000, 4, 1), (48000, 2, 1),
                                             (11025, 0.1, 1), (96000, 2, 1), (5512, 12, 1)]):
        n = int(rate * seconds)
        x = (rng.standard_normal((n, ch)) * 3000 + 8000 * np.sin(np.arange(n)[:, None] * (0.01 + 0.003 * k))).astype(np.int16)
        p = str(tmp_path / f"f{k}.wav")
        _wav(p, x, rate, ch)
        paths.append(p)
        kinds.append("ok")
    paths.insert(2, str(tmp_path / "missing.wav")); kinds.insert(2, "missing")
    bad = str(tmp_path / "bad.caf")
    open(bad, "wb").write(b"caff" + bytes(100))
    paths.insert(5, bad); kinds.insert(5, "bad")
    paths.append(os.path.join(BIRDS, "Crow.caf")); kinds.append("ok")
    for settings in (dict(), dict(sample_rate=8000, window=1024, stride=32, bands=24, subfp_len=120)):
        det = lb.Detective().configure(**settings)
        cfg = oracle.Config(**settings)
        for hop_mode, tail_mode, res in ((1, 1, 0), (1, 2, 1), (0, 1, 2)):
            det.set_file_hop_mode(hop_mode).set_file_tail_mode(tail_mode).set_resampler_mode(res)
            fps, sts = det.process_audio_urls(paths, statuses=True)
            for p, kind, fp, st in zip(paths, kinds, fps, sts):
                if kind == "missing":
                    assert st == -43 and fp is None
                elif kind == "bad":
                    assert st == lb.constant("kLBAudioDetectiveUnsupportedFile") and fp is None
                else:
                    want = oracle.fingerprint_file(p, cfg, hop_mode, tail_mode, res)
                    assert st == 0 and np.array_equal(fp.to_bools().reshape(want.shape), want), (p, settings, hop_mode, tail_mode, res)
        with pytest.raises(lb.LBAudioDetectiveError):
            det.process_audio_urls(paths)                                     # without statuses the first failure is the call's
    assert lb.Detective().process_audio_urls([]) == []


def test_compare_audio_urls_is_one_batch_of_two(lb, gpu, oracle):
    a, b = os.path.join(BIRDS, "BlackBird.caf"), os.path.join(BIRDS, "BlackBird_eql.caf")
    det = lb.Detective()
    cfg = oracle.Config()
    want = oracle.compare_fp(oracle.fingerprint_file(a, cfg), oracle.fingerprint_file(b, cfg), 200)
    got = det.compare_audio_urls(a, b)
    assert np.float32(got).view(np.uint32) == np.float32(want).view(np.uint32) and 0.92 < got < 0.94
    with pytest.raises(lb.LBAudioDetectiveError) as e:
        det.compare_audio_urls(a, "/nonexistent.caf")                         # the SECOND file's status (D.m:449-456)
    assert e.value.status == -43


def test_batch_larger_than_one_pinned_run(lb, gpu, tmp_path):
    """More than 512 MB of files in one call: the batch goes through in RUNS of the pinned block (api_files.cpp), each read
    by the pooled readers, uploaded in parts and processed before the next is read.  Files on both sides of the run
    boundary must equal their single-file results."""
    rng = np.random.default_rng(77)
    contents = []
    for k in range(3):
        n = 8_400_000 + 1000 * k                                            # ~16.8 MB of int16 each
        x = (rng.standard_normal(n) * 2500 + 7000 * np.sin(np.arange(n) * (0.02 + 0.004 * k))).astype(np.int16)
        p = str(tmp_path / f"big{k}.wav")
        _wav(p, x, 11025)
        contents.append(p)
    paths = []
    for i in range(33):                                                     # 33 x 16.8 MB = 554 MB: two runs
        q = str(tmp_path / f"f{i:02d}.wav")
        os.link(contents[i % 3], q)
        paths.append(q)
    det = lb.Detective()
    fps, sts = det.process_audio_urls(paths, statuses=True)
    assert sts == [0] * len(paths)
    singles = {i: det.process_audio_url(paths[i]) for i in (0, 1, 2)}
    for i, fp in enumerate(fps):
        assert fp.number_of_subfingerprints > 0 and fp.equal_to_fingerprint(singles[i % 3]), i


def test_file_batches_from_several_threads(lb, gpu, oracle):
    """The reader pool is shared by every detective of the process and a detective may be called from several threads:
    three threads (two on ONE detective, one on its own) run batch calls at the same time; every result is the
    oracle's."""
    import threading
    paths = _all_birds()[:24]
    cfg = oracle.Config()
    want = [oracle.fingerprint_file(p, cfg) for p in paths[:6]]
    shared, own = lb.Detective(), lb.Detective()
    errors = []

    def worker(det, first):
        try:
            for it in range(6):
                sub = paths[first:] + paths[:first]
                fps = det.process_audio_urls(sub)
                for i in range(6):
                    k = (i - first) % len(paths)
                    got = fps[k].to_bools()
                    assert got.shape == want[i].shape and np.array_equal(got, want[i]), (first, it, i)
        except BaseException as e:                                    # noqa: BLE001 -- reported by the main thread
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(d, f)) for d, f in ((shared, 0), (shared, 5), (own, 11))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_single_file_calls_from_two_threads_on_one_detective(lb, gpu, oracle):
    """Round-3 advice: LBAudioDetectiveProcessAudioURL -- upstream's main entry point -- did not take the detective's lock
    while the batch and pair calls did; two threads on ONE detective raced on the pinned block, the converter buffers and
    the stride.  Two threads now hammer it with different files, a third mixes in CompareAudioURLs; every fingerprint is
    the oracle's."""
    import threading
    paths = _all_birds()[:10]
    cfg = oracle.Config()
    want = {p: oracle.fingerprint_file(p, cfg) for p in paths}
    det = lb.Detective()
    errors = []

    def worker(mine):
        try:
            for _ in range(8):
                for p in mine:
                    got = det.process_audio_url(p).to_bools()
                    assert got.shape == want[p].shape and np.array_equal(got, want[p]), p
        except BaseException as e:                                    # noqa: BLE001
            errors.append(e)

    def comparer():
        try:
            first = det.compare_audio_urls(paths[0], paths[1])
            for _ in range(20):
                assert det.compare_audio_urls(paths[0], paths[1]) == first
        except BaseException as e:                                    # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(paths[:5],)), threading.Thread(target=worker, args=(paths[5:],)),
               threading.Thread(target=comparer)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_many_files_in_one_call_run_as_a_pipeline_with_the_same_results(lb, gpu, oracle):
    """1200 files in one call (the sixty fixtures x 20: 260 MB, eight runs, two in flight -- the host reads run i + 1 while
    the device works on run i) against the same call one run at a time and against the oracle; a missing file and an
    unreadable one in the middle keep their statuses and do not disturb their neighbours."""
    paths = _all_birds()
    cfg = oracle.Config()
    want = [oracle.fingerprint_file(p, cfg) for p in paths]
    batch = paths * 20
    batch[137] = "/nonexistent/file.caf"
    batch[701] = os.path.join(os.path.dirname(BIRDS), "essay_figures.json")        # a file that is no audio file
    det = lb.Detective()
    outs = []
    for pipe in (True, False):
        det.set_file_pipeline(pipe)
        fps, statuses = det.process_audio_urls(batch, statuses=True)
        assert statuses[137] == -43 and statuses[701] != 0 and sum(1 for s in statuses if s != 0) == 2
        outs.append([None if f is None else f.to_bools() for f in fps])
    for k, (a, b) in enumerate(zip(*outs)):
        if k in (137, 701):
            assert a is None and b is None
            continue
        w = want[k % len(paths)]
        assert a.shape == w.shape and np.array_equal(a, w) and np.array_equal(b, w), k
