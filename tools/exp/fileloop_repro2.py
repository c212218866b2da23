"""Stage-2 half of a failing file-loop case: the oracle's frame rows through the device stage 2, against the
oracle's own stage 2 (rows_to_subfingerprints) and a numpy ranking of the device's Haar output."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import lbaudiodetective_amd as lb
from oracle import oracle as O

rate, window, stride, bands, subfp, hop, n_client, file_frames, tail_mode = (8000.0, 128, 100, 24, 235, 50, 553, 38149, 1)
cfg = O.Config(rate, window, stride, bands, 1)
cfg.subfp_len = subfp
x = O.synth_clip(13, 5, 44100, n_client)
want, raw, nread = O.fingerprint_file_loop(x, file_frames, hop, cfg, tail_mode, taps=True)
det = lb.Detective().configure(sample_rate=rate, window=window, stride=stride, bands=bands, subfp_len=subfp)
frames = torch.from_numpy(np.ascontiguousarray(raw, dtype=np.float32)).cuda()
packed, haar = lb.frames_to_subfingerprints_device(det, frames, want_haar=True)
bits = lb.unpack_packed(packed.cpu().numpy(), subfp).reshape(-1, subfp)
print("device stage 2 on the oracle's rows == oracle bits:", np.array_equal(bits, want))
oh = O.rows_to_subfingerprints(raw.reshape(-1, bands), cfg) if hasattr(O, "rows_to_subfingerprints") else None
if oh is not None:
    print("oracle rows_to_subfingerprints == file-loop bits:", np.array_equal(oh.reshape(want.shape), want))
h = haar.cpu().numpy().reshape(raw.shape[0], -1)
for f in range(h.shape[0]):
    hv = h[f]
    key = hv.view(np.uint32) & 0x7fffffff
    order = np.lexsort((np.arange(hv.size), -key.astype(np.int64)))
    keep = (subfp + 1) // 2
    ref = np.zeros(2 * keep, np.uint8)
    for rnk, i in enumerate(order[:keep]):
        if hv[i] > 0: ref[2 * rnk] = 1
        elif hv[i] < 0: ref[2 * rnk + 1] = 1
    ref = ref[:subfp]
    print("frame", f, "numpy ranking of the device Haar == device bits:", np.array_equal(ref, bits[f]), "== oracle bits:", np.array_equal(ref, want[f]),
          "NaN", int(np.isnan(hv).sum()), "zeros", int((hv == 0).sum()), "distinct keys among top 200:", len(np.unique(key[order[:200]])))
    if not np.array_equal(bits[f], want[f]):
        d = np.nonzero(bits[f] != want[f])[0]
        print("   first differing bits", d[:10], "keys around rank", d[0] // 2, [hex(int(k)) for k in key[order[d[0] // 2 - 2: d[0] // 2 + 6]]], order[d[0] // 2 - 2: d[0] // 2 + 6])
