"""Phase times inside compare_sliding_kernel (a -DLBAD_SLIDE_PROF build: LBAD_LIB=...): shader-clock ticks summed over the
waves, per phase: refill, task set-up, pass (of which: ring fill), result."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import lbaudiodetective_amd as lb
from lbaudiodetective_amd import _native as _N
if os.environ.get("LBAD_LIB"):
    _N.LIB_PATH = os.path.abspath(os.environ["LBAD_LIB"])
from oracle import oracle as O
SEED = 0x4C424145
n = 1_000_000
counts = O.synth_ragged_counts(SEED, 0, n, 20, 70)
total = int(counts.sum())
packed = lb.synth_ragged_corpus_device(SEED, 0, counts, 200)
c = lb.Corpus.ragged(200, n, total)
c.append_ragged_packed_device(packed, counts)
c.set_bound_pruning(False)
del packed
L = C.CDLL(_N.LIB_PATH)
key = torch.zeros(1, dtype=torch.int64, device="cuda")
for nq in [int(v) for v in sys.argv[1:]] or [21]:
    src = O.synth_entry(SEED, 777_777, max(int(counts[777_777]), nq), 200)
    q = lb.Fingerprint.from_bools(src[:nq])
    for _ in range(2):
        c.query_key_device(q, key)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    L.LBAudioDetectiveDebugSlideProfile(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    c.query_key_device(q, key)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    L.LBAudioDetectiveDebugSlideProfile(buf, 0)
    v = list(buf)
    passes, waves = v[5], v[8]
    names = ["refill", "task set-up", "pass", "result", "ring fill (inside pass)"]
    print(json.dumps({"nq": nq, "scan_ms": round(ms, 4), "passes": passes, "waves": waves, "shader_GHz": round((v[6] + v[7]) / max(v[9], 1) * 0.1, 3), "vm_wait_ticks_per_A_step": round(v[13] / max(v[14], 1), 1), "wave_busy_ms": round(v[9] / waves * 1e-5, 4), "wave_busy_max_ms": round(v[10] * 1e-5, 4), "wave_busy_min_ms": round(v[11] * 1e-5, 4), "ticks_per_wave_A": round(v[6] / waves), "ticks_per_wave_B": round(v[7] / waves),
                      **{f"{names[i]} ticks/pass": round(v[i] / max(passes, 1)) for i in range(5)}}))
