import torch, numpy as np, collections
import lbaudiodetective_amd as lb
det = lb.Detective().configure(sample_rate=44100.0, window=1024, stride=64)
n = 20000
clips = torch.empty((n, 44100), dtype=torch.float32, device="cuda")
lb.synth_clips_device(0x4C424144, 0, n, 44100, 44100, out=clips)
for _ in range(2):
    out, raw, haar = det.fingerprint_clips_device(clips, taps=True)
torch.cuda.synchronize()
r = raw.cpu().numpy().reshape(n * 5, 4, 4, 8, 32)[:, :, :, 0, :10]   # frame, quarter, wave, 9 values
r = r.reshape(-1, 4, 10)     # unit, wave, vals
d = r[:, :, :6]
print("units", r.shape[0])
names = ["wait+barrier", "points+barrier2+issue", "FFT", "pass 1", "pass 2", "bands"]
mid = d[d.shape[0] // 4: 3 * d.shape[0] // 4]
for i, nm in enumerate(names):
    v = mid[:, :, i].ravel()
    print(f"{nm:28s} mean {v.mean():8.0f} p10 {np.percentile(v,10):8.0f} p50 {np.percentile(v,50):8.0f} p90 {np.percentile(v,90):8.0f}")
tot = mid.sum(axis=2).ravel()
print("total per wave mean", tot.mean(), "p50", np.percentile(tot, 50))
rr = r[r.shape[0]//4:3*r.shape[0]//4, :, 9].ravel()
print("realtime ticks (100 MHz) per iteration mean", rr.mean(), "-> shader clock GHz", tot.mean() / (rr.mean() * 10.0))
# phase alignment: for units in the middle, start time and hw id
t0 = r[:, 0, 6].astype(np.int64) + (r[:, 0, 8].astype(np.int64) << 24)
hw = r[:, 0, 7].astype(np.int64)
wave_id = hw & 15; simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
print("wave_id hist", collections.Counter(wave_id.tolist()).most_common(8))
print("simd hist", collections.Counter(simd.tolist()).most_common(8))
# group by (xcd?, se, sh, cu) can't know xcd; use unit index -> xcd = unit // units_per_xcd
nunits = r.shape[0]; upx = (nunits + 7) // 8
xcd = np.arange(nunits) // upx
key = xcd * 4096 + se * 512 + sh * 256 + cu * 16 + simd
order = np.lexsort((t0, key))
k = key[order]; t = t0[order]; tt = tot.mean()
same = k[1:] == k[:-1]
gaps = (t[1:] - t[:-1])[same]
print("start-to-start gap of consecutive wave-0 starts on the same SIMD: p10/p50/p90", np.percentile(gaps, [10, 50, 90]))
print("mean total per wave-0", r[:, 0, :6].sum(axis=1).mean())
# persistent kernel: per-workgroup sums (static partition: kernel time = slowest workgroup)
wgx = 64
u = np.arange(nunits)
x = u // upx
j = (u - x * upx) % wgx
wg = x * wgx + j
per_unit = r[:, :, :6].sum(axis=2).max(axis=1)      # slowest wave of the unit
tot_wg = np.bincount(wg, weights=per_unit)
print("per-WG total ticks: min/median/max", tot_wg.min(), np.median(tot_wg), tot_wg.max())
for xc in range(8):
    t = tot_wg[xc * wgx:(xc + 1) * wgx]
    print("xcd", xc, "min/med/max", int(t.min()), int(np.median(t)), int(t.max()))
cuid = (se * 2 + sh) * 16 + cu
first_unit_of_wg = {}
print("distinct (xcd, se, sh, cu):", len(set(zip(x.tolist(), se.tolist(), sh.tolist(), cu.tolist()))))
# how many WGs per CU
import collections as C
wg_cu = {}
for w_, xx, c_ in zip(wg.tolist()[::1], x.tolist(), cuid.tolist()):
    wg_cu[w_] = (xx, c_)
cnt = C.Counter(wg_cu.values())
print("WGs per CU histogram:", C.Counter(cnt.values()))
slow = np.argsort(tot_wg)[-8:]
print("slowest WGs:", [(int(w_), wg_cu[int(w_)], int(tot_wg[w_]), cnt[wg_cu[int(w_)]]) for w_ in slow])
fast = np.argsort(tot_wg)[:4]
print("fastest WGs:", [(int(w_), wg_cu[int(w_)], int(tot_wg[w_]), cnt[wg_cu[int(w_)]]) for w_ in fast])
