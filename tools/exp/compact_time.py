"""configs[1] pass with compact inter-stage frames (default) against full rows (variant 4): stage times from the library's
own events, interleaved A/B on one box."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lbaudiodetective_amd import _native as _N
if os.environ.get("LBAD_LIB"):
    _N.LIB_PATH = os.path.abspath(os.environ["LBAD_LIB"])
import lbaudiodetective_amd as lb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
clips = lb.synth_clips_device(0x4C424144, 0, n, 44100, 44100)
dets = {}
for name, variant in (("compact", 0), ("full_rows", 4)):
    d = lb.Detective().configure(sample_rate=44100, window=1024)
    d.set_kernel_variant(variant)
    dets[name] = (d, d.fingerprint_clips_device(clips))
torch.cuda.synchronize()
assert torch.equal(dets["compact"][1], dets["full_rows"][1])
out = {}
for rnd in range(3):
    for name, (d, packed) in dets.items():
        d.set_stage_timing(True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            d.fingerprint_clips_device(clips, out=packed)
        e1.record()
        torch.cuda.synchronize()
        s1, s2, ln = d.stage_times()
        d.set_stage_timing(False)
        out.setdefault(name, []).append({"pass_ms": round(e0.elapsed_time(e1) / 5, 3), "stage1_ms": round(s1 / 5, 3),
                                         "stage2_ms": round(s2 / 5, 3), "stage2_us_per_launch": round(s2 / ln * 1e3, 1)})
print(json.dumps(out, indent=1))
