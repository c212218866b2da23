#!/usr/bin/env python3
"""Throughput of the batch path at the three processing configurations of BASELINE.json, plus the
host-buffer (PCIe-inclusive) rate of configuration B.  Prints one JSON object."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lbaudiodetective_amd as lb

out = {}
for name, rate, window, samples, n, stereo in [("A_5512_2048", 5512, 2048, 5512 * 9, 20000, False),
                                                ("B_44100_1024", 44100, 1024, 44100, 100000, False),
                                                ("C_48000_4096_stereo", 48000, 4096, 48000, 10000, True)]:
    det = lb.Detective().configure(sample_rate=rate, window=window)
    clips = lb.synth_clips_device(0x4C424144, 0, n, rate, samples, stereo)
    packed = det.fingerprint_clips_device(clips)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 3
    for _ in range(reps):
        det.fingerprint_clips_device(clips, out=packed)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    out[name] = {"clips": n, "seconds_per_clip": samples / rate, "subfingerprints_per_clip": int(packed.shape[1]),
                 "ms": round(ms, 3), "audio_seconds_per_s": round(n * samples / rate / (ms * 1e-3), 1),
                 "windows_per_s": round(n * packed.shape[1] * 128 / (ms * 1e-3), 1)}
    del clips, packed
    torch.cuda.empty_cache()

# strides other than 64 (round 3: k_rows_full.hip at any even stride): generic stage-1 kernel against the specialised one,
# same batch, stage-1 time from the library's own events
for name, rate, window, stride, seconds, n in [("hop8_5512_2048", 5512, 2048, 8, 9, 2500), ("stride32_11025_1024", 11025, 1024, 32, 4, 20000),
                                               ("stride128_22050_2048", 22050, 2048, 128, 8, 20000), ("stride16_5512_2048", 5512, 2048, 16, 9, 5000)]:
    samples = int(rate * seconds)
    clips = lb.synth_clips_device(0x4C424144, 0, n, rate, samples)
    row = {"clips": n, "seconds_per_clip": seconds}
    for label, variant in (("generic", 1), ("specialised", 2)):
        det = lb.Detective().configure(sample_rate=rate, window=window, stride=stride)
        det.set_kernel_variant(variant)
        packed = det.fingerprint_clips_device(clips)
        torch.cuda.synchronize()
        det.set_stage_timing(True)
        for _ in range(3):
            det.fingerprint_clips_device(clips, out=packed)
        s1, s2, ln = det.stage_times()
        row[label + "_stage1_ms"] = round(s1 / 3, 3)            # per pass (three passes timed; a pass may be several chunks)
        row["windows"] = int(n * packed.shape[1] * 128)
        row[label + "_bits_sum"] = int(packed.to(torch.int64).sum().item())
    row["speedup"] = round(row["generic_stage1_ms"] / row["specialised_stage1_ms"], 2)
    row["identical"] = row.pop("generic_bits_sum") == row.pop("specialised_bits_sum")
    out[name] = row
    del clips, packed
    torch.cuda.empty_cache()

# band counts other than 32 on the register Haar / select kernel (round 3): stage-2 time, generic against specialised
for name, rate, window, bands, subfp, n in [("bands16_16000_1024", 16000, 1024, 16, 200, 100000), ("bands64_22050_2048", 22050, 2048, 64, 256, 50000),
                                            ("bands32_44100_1024", 44100, 1024, 32, 200, 100000)]:
    samples = window + 64 * 128 * 5
    clips = lb.synth_clips_device(0x4C424144, 0, n, rate, samples)
    row = {"clips": n, "frames": n * 5}
    for label, variant in (("generic", 1), ("specialised", 0)):
        det = lb.Detective().configure(sample_rate=rate, window=window, bands=bands, subfp_len=subfp)
        det.set_kernel_variant(variant)
        packed = det.fingerprint_clips_device(clips)
        torch.cuda.synchronize()
        det.set_stage_timing(True)
        for _ in range(3):
            det.fingerprint_clips_device(clips, out=packed)
        s1, s2, ln = det.stage_times()
        row[label + "_stage2_ms"] = round(s2 / 3, 3)
        row[label + "_bits_sum"] = int(packed.to(torch.int64).sum().item())
    row["speedup"] = round(row["generic_stage2_ms"] / row["specialised_stage2_ms"], 2)
    row["identical"] = row.pop("generic_bits_sum") == row.pop("specialised_bits_sum")
    out[name] = row
    del clips, packed
    torch.cuda.empty_cache()

# host buffers in, Booleans out (H2D of the PCM and D2H of the packed bits inside the timed region)
det = lb.Detective().configure(sample_rate=44100, window=1024)
host = lb.synth_clips_device(0x4C424144, 0, 4000, 44100, 44100).cpu().numpy()
det.fingerprint_clips(host[:100])
t0 = time.perf_counter()
det.fingerprint_clips(host)
dt = time.perf_counter() - t0
out["B_host_buffers_pageable"] = {"clips": 4000, "s": round(dt, 4), "audio_seconds_per_s": round(4000 / dt, 1)}
host16 = np.round(host * 32768).astype(np.int16)
t0 = time.perf_counter()
det.fingerprint_clips(host16)
dt = time.perf_counter() - t0
out["B_host_buffers_pageable_int16"] = {"clips": 4000, "s": round(dt, 4), "audio_seconds_per_s": round(4000 / dt, 1)}
pinned = torch.from_numpy(host).pin_memory()
dev = torch.empty((4000, 44100), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
t0 = time.perf_counter()
dev.copy_(pinned, non_blocking=True)
p = det.fingerprint_clips_device(dev)
res = p.cpu()
dt = time.perf_counter() - t0
out["B_host_buffers_pinned"] = {"clips": 4000, "s": round(dt, 4), "audio_seconds_per_s": round(4000 / dt, 1),
                                "h2d_GBps": round(host.nbytes / dt / 1e9, 2)}
print(json.dumps(out))
