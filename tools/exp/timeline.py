import torch, numpy as np, collections
import lbaudiodetective_amd as lb
det = lb.Detective().configure(sample_rate=44100.0, window=1024, stride=64)
n = 20000
clips = torch.empty((n, 44100), dtype=torch.float32, device="cuda")
lb.synth_clips_device(0x4C424144, 0, n, 44100, 44100, out=clips)
for _ in range(2):
    out, raw, haar = det.fingerprint_clips_device(clips, taps=True)
torch.cuda.synchronize()
r = raw.cpu().numpy().reshape(n * 5, 4, 4, 8, 32)[:, :, :, 0, :9]   # frame, quarter, wave, 9 values
r = r.reshape(-1, 4, 9)     # unit, wave, vals
d = r[:, :, :6]
print("units", r.shape[0])
names = ["load issue", "barrier wait", "points+FFT", "pass1 rows+trees", "pass2 rows+trees+split", "bands+store"]
mid = d[d.shape[0] // 4: 3 * d.shape[0] // 4]
for i, nm in enumerate(names):
    v = mid[:, :, i].ravel()
    print(f"{nm:28s} mean {v.mean():8.0f} p10 {np.percentile(v,10):8.0f} p50 {np.percentile(v,50):8.0f} p90 {np.percentile(v,90):8.0f}")
tot = mid.sum(axis=2).ravel()
print("total per wave mean", tot.mean(), "p50", np.percentile(tot, 50))
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
