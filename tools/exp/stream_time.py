"""Stage-1 time of configs[4] (10 000 stereo-summed 1 s clips, 48 kHz / 4096) for a given library build:
    python tools/exp/stream_time.py [path/to/lib.so ...]"""
import sys, torch
import lbaudiodetective_amd._native as N
import subprocess, os, json
child = "--child" in sys.argv
cfg_a = "--A" in sys.argv                                   # 5512 Hz / 2048, 20 000 nine-second clips instead
cfg_b = "--B" in sys.argv                                   # 44.1 kHz / 1024, 100 000 one-second clips (the bench workload)
libs = [a for a in sys.argv[1:] if a not in ("--child", "--A", "--B")] or [N.LIB_PATH]
if not child:
    for l in libs:
        out = subprocess.run([sys.executable, __file__, l, "--child"] + (["--A"] if cfg_a else []) + (["--B"] if cfg_b else []), capture_output=True, text=True, timeout=120)
        print(os.path.basename(l), out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
    sys.exit(0)
N.LIB_PATH = libs[0]
import lbaudiodetective_amd as lb
if cfg_b:
    det = lb.Detective().configure(sample_rate=44100, window=1024)
    big = lb.synth_clips_device(0x4C424144, 0, 100000, 44100, 44100, False)
elif cfg_a:
    det = lb.Detective().configure(sample_rate=5512, window=2048)
    big = lb.synth_clips_device(0x4C424144, 0, 20000, 5512, 5512 * 9, False)
else:
    det = lb.Detective().configure(sample_rate=48000, window=4096)
    big = lb.synth_clips_device(0x4C424144, 0, 10000, 48000, 48000, True)
out = det.fingerprint_clips_device(big)
torch.cuda.synchronize()
det.set_stage_timing(True)
for _ in range(5):
    det.fingerprint_clips_device(big, out=out)
s1, s2, launches = det.stage_times()
print(json.dumps({"stage1_ms": round(s1 / launches, 3), "stage2_ms": round(s2 / launches, 3)}))
