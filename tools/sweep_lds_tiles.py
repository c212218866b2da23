#!/usr/bin/env python3
"""BASELINE configs[4]: 48 kHz / 4096-point windows, stereo-summed clips -- LDS-tile sizing sweep of the
generic stage-1 kernel: waves per workgroup x twiddle cache on/off (LBAudioDetectiveSetKernelTuning).  Each
point runs in a fresh process so that it can be profiled on its own.  Prints a markdown table.

    python tools/sweep_lds_tiles.py                       # the sweep
    python tools/sweep_lds_tiles.py --one WAVES CACHE     # one point (what the rocprofv3 passes run)
"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if "--one" in sys.argv:
    import torch
    import lbaudiodetective_amd as lb
    n = 10000
    i = sys.argv.index("--one")
    det = lb.Detective().configure(sample_rate=48000, window=4096)
    det.set_kernel_variant(1)                      # the generic kernel is the one being swept
    det.set_kernel_tuning(int(sys.argv[i + 1]), bool(int(sys.argv[i + 2])))
    clips = lb.synth_clips_device(0x4C424144, 0, n, 48000, 48000, True)
    out = det.fingerprint_clips_device(clips)
    torch.cuda.synchronize()
    det.set_stage_timing(True)
    for _ in range(3):
        det.fingerprint_clips_device(clips, out=out)
    s1, s2, launches = det.stage_times()
    ms = s1 / launches
    windows = n * out.shape[1] * 128
    print(json.dumps({"stage1_ms": ms, "stage2_ms": s2 / launches, "windows_per_s": windows / (ms * 1e-3),
                      "pcm_GBps": n * 48000 * 4 / (ms * 1e-3) / 1e9}))
    sys.exit(0)

rows = []
CACHE = 28672            # per-lane twiddle cache of passes 1..2 at W = 4096 (pass 0: five stages on scalar constants)
NREAD = 384              # bins 3..347 read by the bands, padded to 64
PER_WAVE = (2 * (2048 + 64) + NREAD) * 4
for wpb in (1, 2, 4, 6, 7):
    for nocache in (0, 1):
        out = subprocess.run([sys.executable, __file__, "--one", str(wpb), str(1 - nocache)], capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        r = json.loads(line[-1]) if line else {"stage1_ms": float("nan"), "windows_per_s": 0, "pcm_GBps": 0}
        lds = wpb * PER_WAVE + 2 * NREAD * 4 + (0 if nocache else CACHE)
        per_cu = min(160 * 1024 // lds, 32 // wpb)
        rows.append((wpb, "off" if nocache else "on", lds, per_cu, per_cu * wpb, r["stage1_ms"], r["windows_per_s"], r["pcm_GBps"]))
print("| waves / workgroup | twiddle cache | LDS per workgroup (B) | workgroups / CU | waves / CU | stage-1 ms (10 000 clips) | windows/s | algorithmic PCM GB/s |")
print("|---|---|---|---|---|---|---|---|")
for wpb, c, lds, per_cu, waves, ms, wps, gb in rows:
    print(f"| {wpb} | {c} | {lds} | {per_cu} | {waves} | {ms:.2f} | {wps:.3g} | {gb:.1f} |")
