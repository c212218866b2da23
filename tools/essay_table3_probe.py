#!/usr/bin/env python3
"""The one Boolean-level output the reference publishes: the essay's Table 3 (PDF p.36, "The database entry of the
Blackbird") prints `fingerprint 10010101001010...` -- fourteen Booleans of a stored fingerprint.  CPU only.

Compared here against the oracle's fingerprints of ALL ten archive recordings under every model of
tools/birds_matrix.py --sweep (3 converters x 2 IMA4 packet starts x 3 end-of-file treatments, both hop modes), in

  rank order      what LBAudioDetectiveFrame.m:165-191 (and the essay's appendix, p.60) writes: the sign pair of the
                  i-th LARGEST coefficient at Booleans 2 i, 2 i + 1
  position order  what the Waveprint scheme the essay describes (p.26-28) writes and an older build may have stored:
                  coefficient (row, column) at Booleans 2 (row * 32 + column) .., zero pairs for everything outside the top 200

A `00` pair in fifth place cannot come out of the rank order unless a coefficient ranked fifth is exactly zero (or NaN);
in position order it only says that the frame's fifth coefficient is not among the 200 largest.
    python tools/essay_table3_probe.py [out.json]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O

PRINTED = "10010101001010"
BIRDS = ["BlackBird", "BlueTit", "Chaffinch", "Chiffchaff", "Crow", "GreatTit", "Kestrel", "Pigeon", "Sparrow", "Wren"]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def frames_of(path, cfg, hop_mode, tail, resampler):
    """Haar-decomposed frames [n, 128, 32] of a file through the oracle (decode, convert, upstream's window loop)."""
    x, rate = O.decode_audio_file(path)
    client = O.resample(x, rate, cfg.sample_rate, resampler)
    if hop_mode == 0:
        _, raw = O.fingerprint_pcm(client, cfg, taps=True)[:2]
    else:
        hop = max(1, int(np.floor(cfg.stride * cfg.sample_rate / rate + 0.5)))
        _, raw, _ = O.fingerprint_file_loop(client, x.size, hop, cfg, tail, taps=True)
    return np.stack([O.haar_2d(f) for f in raw]) if len(raw) else np.zeros((0, 128, 32), np.float32)


def rank_order(frame, n=7):
    return "".join("".join(str(int(b)) for b in O.extract(frame, 200)[: 2 * n]))


def position_order(frame, n=7, keep=200):
    flat = frame.reshape(-1)
    order = np.lexsort((np.arange(flat.size), -np.abs(flat.astype(np.float64))))      # |v| descending, lower index first
    top = set(order[:keep].tolist())
    s = ""
    for p in range(n):
        v = flat[p]
        s += "10" if (p in top and v > 0) else ("01" if (p in top and v < 0) else "00")
    return s


def main():
    cfg = O.Config()
    rows, hits = [], []
    for carry in (0, 1):
        O.lib().lbo_file_set_ima4_carry(carry)
        for resampler in (0, 1, 2):
            for hop_mode, tail in ((1, 1), (1, 2), (1, 0), (0, 0)):
                for bird in BIRDS:
                    fr = frames_of(os.path.join(ROOT, "tests", "golden", "birds", bird + ".caf"), cfg, hop_mode, tail, resampler)
                    for i, f in enumerate(fr):
                        r, p = rank_order(f), position_order(f)
                        for kind, s in (("rank", r), ("position", p)):
                            agree = sum(a == b for a, b in zip(s, PRINTED))
                            if agree == len(PRINTED):
                                hits.append({"bird": bird, "subfingerprint": i, "order": kind, "ima4": carry, "resampler": resampler,
                                             "hop_mode": hop_mode, "tail": tail})
                        if i == 0:
                            rows.append({"bird": bird, "ima4_start": "carry" if carry else "header", "resampler": resampler,
                                         "hop_mode": hop_mode, "tail": tail, "rank_order": r, "position_order": p,
                                         "rank_agree": sum(a == b for a, b in zip(r, PRINTED)),
                                         "position_agree": sum(a == b for a, b in zip(p, PRINTED)),
                                         "zero_pairs_in_rank_order": r.count("00") if len(r) % 2 == 0 and all(r[j:j + 2] != "11" for j in range(0, 14, 2)) else None})
    O.lib().lbo_file_set_ima4_carry(0)
    by_first = {}
    for r in rows:
        by_first.setdefault((r["bird"], r["rank_order"], r["position_order"]), 0)
        by_first[(r["bird"], r["rank_order"], r["position_order"])] += 1
    out = {"printed": PRINTED, "models": len(rows) // len(BIRDS), "exact_matches_anywhere": hits,
           "best_rank_agreement_first_subfingerprint": max(r["rank_agree"] for r in rows),
           "best_position_agreement_first_subfingerprint": max(r["position_agree"] for r in rows),
           "blackbird_first_subfingerprint": sorted({(r["rank_order"], r["position_order"]) for r in rows if r["bird"] == "BlackBird"}),
           "first_subfingerprints": rows}
    print(json.dumps({k: out[k] for k in out if k != "first_subfingerprints"}, indent=1))
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
