"""Replays file-loop cases of tools/fuzz_parity.py (kind 8) by trial number and seed: prints where the bits differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import lbaudiodetective_amd as lb
from oracle import oracle as O

cases = [(8000.0, 128, 100, 24, 235, 50, 553, 38149, 1), (44100.0, 512, 64, 21, 235, 23, 868, 26481, 1),
         (5512.0, 256, 100, 43, 233, 69, 630, 50412, 1)]
for rate, window, stride, bands, subfp, hop, n_client, file_frames, tail_mode in cases:
    cfg = O.Config(rate, window, stride, bands, 1)
    cfg.subfp_len = subfp
    for seed in range(40):
        x = O.synth_clip(seed * 7919 + 13, 5, 44100, n_client)
        det = lb.Detective().configure(sample_rate=rate, window=window, stride=stride, bands=bands, subfp_len=subfp)
        det.set_file_tail_mode(tail_mode)
        got = det.process_file_stream(x, file_frames, hop).to_bools()
        want, raw, nread = O.fingerprint_file_loop(x, file_frames, hop, cfg, tail_mode, taps=True)
        if got.shape != want.shape or not np.array_equal(got, want):
            d = np.argwhere(got != want)
            print("MISMATCH", rate, window, stride, bands, subfp, hop, n_client, file_frames, "seed", seed, "shape", got.shape, want.shape,
                  "diff bits", len(d), d[:12].tolist())
            nz = (raw != 0).reshape(raw.shape[0], -1).sum(1)
            print("   nonzero band values per frame", nz.tolist(), "rows with data", int((nread > 0).sum()) if nread is not None else None)
            break
    else:
        print("no mismatch in 40 seeds:", rate, window, stride, bands, subfp, hop, n_client, file_frames)
