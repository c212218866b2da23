"""Per-wave time stamps of compare_short_multi_kernel (a -DLBAD_SLIDE_STAMPS build, LBAD_LIB=...): start, end of the first
chunk, end of chunk 17, end -- do the waves of the scan finish together, and how long is a chunk?"""
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
keys = torch.zeros(8, dtype=torch.int64, device="cuda")
for nq in [int(v) for v in sys.argv[1:]] or [5]:
    fps = []
    for k in range(8):
        e = min(100_000 * (k + 1) + 777, n - 1)
        fps.append(lb.Fingerprint.from_bools(O.synth_entry(SEED, e, max(int(counts[e]), nq), 200)[:nq]))
    for _ in range(3):
        c.query_batch_keys_device(fps, keys)
    torch.cuda.synchronize()
    L.LBAudioDetectiveDebugSlideTimesReset()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    c.query_batch_keys_device(fps, keys)
    e1.record()
    torch.cuda.synchronize()
    tb = (C.c_ulonglong * (256 * 16 * 8))()
    L.LBAudioDetectiveDebugSlideTimes(tb, 256 * 16 * 8)
    t = np.array(list(tb), dtype=np.float64).reshape(-1, 8)
    full = t.copy()
    if full[:, 0].min() > 0:
        life = (full[:, 3] - full[:, 0]) / 100.0
        wg = life.reshape(-1, 4)
        print("life by wave of the workgroup:", [round(float(x), 1) for x in wg.mean(axis=0)])
        print("life by workgroup index mod 8:", [round(float(wg[i::8].mean()), 1) for i in range(8)])
        print("life by workgroup index // 256:", [round(float(wg[i * 256:(i + 1) * 256].mean()), 1) for i in range(wg.shape[0] // 256)])
        print("life by (workgroup index // 8) mod 32:", [round(float(np.concatenate([wg[j * 8:(j + 1) * 8] for j in range(i, wg.shape[0] // 8, 32)]).mean()), 1) for i in range(32)])
        print("spread inside a workgroup (max - min), percentiles:", [round(float(x), 1) for x in np.percentile(wg.max(axis=1) - wg.min(axis=1), [10, 50, 90])])
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    us = (t[:, :4] - t0) / 100.0
    pct = lambda v: [round(float(x), 1) for x in np.percentile(v, [0, 10, 50, 90, 100])]
    print(json.dumps({"nq": nq, "ms_events": round(e0.elapsed_time(e1), 4), "waves": int(t.shape[0]),
                      "start_us": pct(us[:, 0]), "first_chunk_us": pct(us[:, 1] - us[:, 0]),
                      "chunks_2_to_17_us_each": pct((us[:, 2] - us[:, 1]) / 16.0), "end_us": pct(us[:, 3]),
                      "life_us": pct(us[:, 3] - us[:, 0])}))
