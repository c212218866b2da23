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

    # per-wave records (100 MHz stamps): start, cursor dry, end of the A part, end, tasks queued when dry, passes after dry, their ticks
    tb = (C.c_ulonglong * (256 * 16 * 8))()
    if hasattr(L, "LBAudioDetectiveDebugSlideTimes"):
        L.LBAudioDetectiveDebugSlideTimesReset()
        c.query_key_device(q, key)
        torch.cuda.synchronize()
        L.LBAudioDetectiveDebugSlideTimes(tb, 256 * 16 * 8)
        t = np.array(list(tb), dtype=np.float64).reshape(256, 16, 8)
        t0 = t[:, :, 0].min()
        us = (t[:, :, :4] - t0) / 100.0
        wg_end = us[:, :, 3].max(1)
        first_dry = np.where(t[:, :, 1] > 0, us[:, :, 1], np.inf).min(1)
        last_dry = us[:, :, 1].max(1)
        queued, late_n, late_ticks = t[:, :, 4], t[:, :, 5], t[:, :, 6]
        print(json.dumps({"nq": nq, "us": {
            "wg_first_wave_dry_mean_min_max": [round(float(v), 1) for v in (first_dry.mean(), first_dry.min(), first_dry.max())],
            "wg_last_wave_dry_mean_min_max": [round(float(v), 1) for v in (last_dry.mean(), last_dry.min(), last_dry.max())],
            "wg_end_mean_min_max": [round(float(v), 1) for v in (wg_end.mean(), wg_end.min(), wg_end.max())],
            "wave_end_percentiles_10_50_90": [round(float(v), 1) for v in np.percentile(us[:, :, 3], [10, 50, 90])],
            "drain_per_wg_mean_max": [round(float((wg_end - first_dry).mean()), 1), round(float((wg_end - first_dry).max()), 1)],
            "wg_end_by_xcd": [round(float(wg_end[x::8].mean()), 1) for x in range(8)],
            "tasks_queued_when_dry_mean_max": [round(float(queued.mean()), 1), float(queued.max())],
            "passes_after_dry_mean_max": [round(float(late_n.mean()), 2), float(late_n.max())],
            "late_pass_us_mean": round(float(late_ticks.sum() / max(late_n.sum(), 1) / (v[6] + v[7]) * v[9] * 0.01), 2),
            "wave_dry_percentiles_10_50_90": [round(float(x), 1) for x in np.percentile(us[:, :, 1][t[:, :, 1] > 0], [10, 50, 90])]}}))

        pb = (C.c_uint * (256 * 16 * 64 * 4))()
        if hasattr(L, "LBAudioDetectiveDebugSlidePasses") and L.LBAudioDetectiveDebugSlidePasses(pb, 256 * 16 * 64 * 4) == 0:
            pp = np.array(list(pb), dtype=np.float64).reshape(256 * 16, 64, 4)
            n_p = t[:, :, 7].reshape(-1).astype(int)
            ghz = (v[6] + v[7]) / max(v[9], 1) * 0.1
            end_us = (pp[:, :, 0] - (t0 % 2**32)) / 100.0          # pass END time stamps
            dur_us = pp[:, :, 1] / ghz / 1e3
            fill_us = (pp[:, :, 2].astype(np.int64) & 0xFFFF) / ghz / 1e3      # vector load of a line in L2
            sload_us = (pp[:, :, 2].astype(np.int64) >> 16) / ghz / 1e3      # scalar load
            probe_us = pp[:, :, 3] / ghz / 1e3
            valid = np.arange(64)[None, :] < np.minimum(n_p, 64)[:, None]
            rows = []
            for b0 in np.arange(0, 600, 40):
                m = valid & (end_us >= b0) & (end_us < b0 + 40)
                if m.sum():
                    rows.append([int(b0), int(m.sum()), round(float(dur_us[m].mean()), 1), round(float(np.percentile(dur_us[m], 90)), 1), round(float(fill_us[m].mean()), 1), round(float(np.percentile(fill_us[m], 90)), 1), round(float(probe_us[m].mean()), 2), round(float(np.percentile(probe_us[m], 90)), 2), round(float(sload_us[m].mean()), 2), round(float(np.percentile(sload_us[m], 90)), 2)])
            print(json.dumps({"passes by END time [t0_us, passes, mean_us, p90_us, fill_mean_us, fill_p90_us, probe_mean_us, probe_p90_us, sload_mean, sload_p90]": rows}))
