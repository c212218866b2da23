import numpy as np, torch
import lbaudiodetective_amd as lb
from oracle import oracle as O
rng = np.random.default_rng(0)
for bands in (14, 7, 12, 32, 20):
    m = np.zeros((128, bands), np.float32)
    m[:, bands - 2] = rng.random(128).astype(np.float32) * 100
    want = O.haar_2d(m.copy())
    det = lb.Detective().configure(sample_rate=22050, window=256, stride=277, bands=bands, subfp_len=20)
    packed, haar = lb.frames_to_subfingerprints_device(det, torch.from_numpy(m[None]).cuda(), want_haar=True)
    got = haar.cpu().numpy()[0]
    d = np.argwhere(got.view(np.uint32) != want.view(np.uint32))
    print("bands", bands, "stage-2 kernel diffs", len(d), d[:4].tolist())
    f = lb.Frame(128)
    for r in range(128):
        f.set_row(m[r], r)
    f.decompose()
    got2 = np.stack([f.get_row(r, bands) for r in range(128)])
    d2 = np.argwhere(got2.view(np.uint32) != want.view(np.uint32))
    print("          Frame.decompose diffs", len(d2), d2[:4].tolist())
