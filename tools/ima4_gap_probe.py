#!/usr/bin/env python3
"""How much of a fingerprint depends on the one IMA4 detail that cannot be read off Apple's decoder: whether a packet
restarts from its 16-bit header (9 bits of predictor; the model of this repository) or carries the running
predictor when the header agrees with it (what e.g. ffmpeg's QuickTime IMA decoder does).  CPU only, oracle only:
all sixty bird fixtures decoded both ways, converted, fingerprinted with upstream's file loop; then upstream's five
tests (tools/birds_matrix.py) on both.

    python tools/ima4_gap_probe.py [--json out.json]
"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from oracle import oracle as O
import birds_matrix as bm

ap = argparse.ArgumentParser()
ap.add_argument("--json")
args = ap.parse_args()
cfg = O.Config()
names = sorted(f[:-4] for f in os.listdir(bm.DIR) if f.endswith(".caf"))


def run(carry):
    O.lib().lbo_file_set_ima4_carry(int(carry))
    pcm, fps = {}, {}
    for n in names:
        p = os.path.join(bm.DIR, n + ".caf")
        pcm[n] = O.decode_audio_file(p)[0]
        fps[n] = O.fingerprint_file(p, cfg, 1, 1, 0)
    O.lib().lbo_file_set_ima4_carry(0)
    return pcm, fps


pcm0, fp0 = run(False)
pcm1, fp1 = run(True)
ima = [n for n in names if not np.array_equal(pcm0[n], pcm1[n])]
rms = {n: float(np.sqrt(np.mean((pcm0[n].astype(np.float64) - pcm1[n]) ** 2))) for n in ima}
sub_total = sum(fp0[n].shape[0] for n in names)
sub_changed = sum(int((fp0[n] != fp1[n]).any(axis=1).sum()) for n in names)
bits_changed = sum(int((fp0[n] != fp1[n]).sum()) for n in names)
out = {"files": len(names), "files_whose_pcm_differs": len(ima),
       "pcm_rms_difference_max": max(rms.values()) if rms else 0.0, "pcm_rms_difference_mean": float(np.mean(list(rms.values()))) if rms else 0.0,
       "subfingerprints": sub_total, "subfingerprints_changed": sub_changed, "booleans_changed": bits_changed,
       "booleans_total": sub_total * cfg.subfp_len}
tests = {}
for t in bm.TESTS:
    s = bm.ESSAY["tests"][t]["suffix"]
    m0 = np.array([[O.compare_fp(fp0[a], fp0[b + s], 200) * 100 for b in bm.BIRDS] for a in bm.BIRDS])
    m1 = np.array([[O.compare_fp(fp1[a], fp1[b + s], 200) * 100 for b in bm.BIRDS] for a in bm.BIRDS])
    tests[t] = {"max_abs_change_of_a_match_percent": round(float(np.abs(m0 - m1).max()), 4),
                "diagonal_header": [round(float(v), 2) for v in np.diag(m0)], "diagonal_carry": [round(float(v), 2) for v in np.diag(m1)],
                "identified_header": int((m0.argmax(axis=1) == np.arange(10)).sum()), "identified_carry": int((m1.argmax(axis=1) == np.arange(10)).sum())}
out["upstream_tests"] = tests
print(json.dumps(out, indent=1))
if args.json:
    json.dump(out, open(args.json, "w"), indent=1)
