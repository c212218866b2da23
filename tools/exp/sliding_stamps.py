"""Per-wave time stamps of compare_sliding_kernel (a -DLBAD_SLIDE_STAMPS build, LBAD_LIB=...): when the waves start, find
their workgroup's cursor empty, and end -- the un-instrumented kernel otherwise (three stores per wave)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import lbaudiodetective_amd as lb
from lbaudiodetective_amd import _native as _N
from oracle import oracle as O
SEED = 0x4C424145
n = int(os.environ.get("N", 1_000_000))
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
    src = O.synth_entry(SEED, min(777_777, n - 1), max(int(counts[min(777_777, n - 1)]), nq), 200)
    q = lb.Fingerprint.from_bools(src[:nq])
    for _ in range(3):
        c.query_key_device(q, key)
    torch.cuda.synchronize()
    L.LBAudioDetectiveDebugSlideTimesReset()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    c.query_key_device(q, key)
    e1.record()
    torch.cuda.synchronize()
    tb = (C.c_ulonglong * (256 * 16 * 8))()
    L.LBAudioDetectiveDebugSlideTimes(tb, 256 * 16 * 8)
    t = np.array(list(tb), dtype=np.float64).reshape(256, 16, 8)
    used = t[:, :, 0] > 0
    t0 = t[:, :, 0][used].min()
    us = (t[:, :, :4] - t0) / 100.0
    end = us[:, :, 3][used]
    dry = us[:, :, 1][used & (t[:, :, 1] > 0)]
    wg_end = np.array([us[g, :, 3][used[g]].max() for g in range(256) if used[g].any()])
    wg_first_dry = np.array([us[g, :, 1][used[g] & (t[g, :, 1] > 0)].min() for g in range(256) if (used[g] & (t[g, :, 1] > 0)).any()])
    print(json.dumps({"nq": nq, "scan_ms_events": round(e0.elapsed_time(e1), 4), "waves": int(used.sum()),
                      "wave_start_us_max": round(float(us[:, :, 0][used].max()), 1),
                      "wave_dry_us_p10_p50_p90": [round(float(v), 1) for v in np.percentile(dry, [10, 50, 90])],
                      "wave_end_us_p10_p50_p90_max": [round(float(v), 1) for v in list(np.percentile(end, [10, 50, 90])) + [end.max()]],
                      "wg_first_dry_us_mean": round(float(wg_first_dry.mean()), 1), "wg_end_us_mean_min_max": [round(float(v), 1) for v in (wg_end.mean(), wg_end.min(), wg_end.max())],
                      "tasks_queued_when_dry_mean": round(float(t[:, :, 4][used].mean()), 1)}))
