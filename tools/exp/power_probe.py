"""Shader clock and board power while one stage-1 kernel runs in a loop:
    python tools/exp/power_probe.py [B|A|C] [seconds]
Samples `rocm-smi --showpower --showclocks --json` (falls back to amdgpu's hwmon files) from a side thread."""
import glob, json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import lbaudiodetective_amd as lb

which = sys.argv[1] if len(sys.argv) > 1 else "B"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
rate, window, n, spc, stereo = {"B": (44100, 1024, 100000, 44100, False), "A": (5512, 2048, 20000, 5512 * 9, False),
                                "C": (48000, 4096, 10000, 48000, True)}[which]
det = lb.Detective().configure(sample_rate=rate, window=window)
clips = lb.synth_clips_device(0x4C424144, 0, n, rate, spc, stereo)
out = det.fingerprint_clips_device(clips)
torch.cuda.synchronize()
samples, stop = [], False


def read_smi():
    try:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5)
        d = json.loads(r.stdout)
        card = next(iter(d.values()))
        return {k: v for k, v in card.items() if "ower" in k or "sclk" in k.lower() or "fclk" in k.lower()}
    except Exception as e:                                  # noqa: BLE001
        vals = {}
        for f in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"):
            try:
                vals[f.split("/")[-1]] = int(open(f).read())
            except OSError:
                pass
        return vals or {"error": repr(e)}


def sampler():
    while not stop:
        samples.append((time.time(), read_smi()))
        time.sleep(0.2)


idle = read_smi()
th = threading.Thread(target=sampler)
th.start()
t0 = time.time()
steps = 0
det.set_stage_timing(True)
while time.time() - t0 < secs:
    for _ in range(10):
        det.fingerprint_clips_device(clips, out=out)
    torch.cuda.synchronize()
    steps += 10
s1, s2, launches = det.stage_times()
stop = True
th.join()
print("idle:", idle)
print(f"{which}: {steps} passes, stage 1 {s1 / launches:.3f} ms, stage 2 {s2 / launches:.3f} ms")
for t, v in samples[:: max(1, len(samples) // 12)]:
    print(f"  t+{t - t0:5.2f} s {v}")
