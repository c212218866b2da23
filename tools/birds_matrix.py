#!/usr/bin/env python3
"""Upstream's own test suite (LBAudioDetectiveTests.m:53-117: Tests 1, 2, 3.1, 3.2, 4) on the sixty bundled
bird fixtures, compared with the numbers the essay publishes for it (tests/golden/essay_figures.json,
Fig. 24-28).  Every original recording is compared with the ten sequences of one suffix.

    python tools/birds_matrix.py [--engine gpu|oracle] [--hop 1] [--tail 1] [--resampler 0] [--check] [--json out.json]

engine gpu     LBAudioDetectiveCompareAudioURLs of the HIP library (the product path)
engine oracle  oracle/lbad_file_oracle.c (container, decode, converter) + oracle/lbad_oracle.c (no GPU, no product code)
--check        exit 1 unless the bounds of `check()` hold (the ones tests/ assert)
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

BIRDS = ["BlackBird", "BlueTit", "Chaffinch", "Sparrow", "GreatTit", "Crow", "Wren", "Chiffchaff", "Kestrel", "Pigeon"]
DIR = os.path.join(ROOT, "tests", "golden", "birds")
ESSAY = json.load(open(os.path.join(ROOT, "tests", "golden", "essay_figures.json")))
TESTS = ["test1", "test2", "test3_1", "test3_2", "test4"]

# Two fixtures cannot reproduce Fig. 24 whatever the converter or tail model (HISTORY.md, rounds 1-3 text, section 8.1):
#   Chaffinch  Chaffinch_eql.caf is the only `_eql` file that is NOT a bit-exact prefix of its original (it was
#              re-encoded: rms error 0.014, 5.7 dB SNR in the 231-2040 Hz band the bands read), so its first 19
#              sub-fingerprints match at ~0.57 instead of 1.0; the essay's 93.0 equals the lossless birds' value.
#   Wren       98.9 % would need 20.8 of 21 sub-fingerprints to agree, but the sequence's last frame lies wholly
#              in the unreadable tail of the file (165 of its last 256 windows), which caps ANY 21-frame
#              sequence at 20/21 = 95.2 %; the nine other birds sit at 92.7-93.6 % for exactly that reason.
UNREACHABLE_TEST1 = {"Chaffinch", "Wren"}


def path(bird, suffix=""):
    return os.path.join(DIR, bird + suffix + ".caf")


def fingerprints_gpu(names, hop, tail, resampler):
    import lbaudiodetective_amd as lb
    det = lb.Detective()
    det.set_file_hop_mode(hop).set_file_tail_mode(tail).set_resampler_mode(resampler)
    return {n: det.process_audio_url(os.path.join(DIR, n + ".caf")).to_bools() for n in names}


def fingerprints_oracle(names, hop, tail, resampler):
    """Entirely the oracle: its own CAF / IMA4 / LPCM reader and converter (oracle/lbad_file_oracle.c), then the
    file loop of oracle/lbad_oracle.c.  Nothing of the product runs."""
    from oracle import oracle as O
    from concurrent.futures import ThreadPoolExecutor
    cfg = O.Config()

    def one(n):
        return O.fingerprint_file(os.path.join(DIR, n + ".caf"), cfg, hop, tail, resampler)
    with ThreadPoolExecutor(min(8, os.cpu_count() or 1)) as ex:
        return dict(zip(names, ex.map(one, names)))


def matrices(engine="gpu", hop=1, tail=1, resampler=0, tests=TESTS):
    """{test: 10 x 10 match matrix in percent}, rows = originals, columns = sequences."""
    from oracle import oracle as O
    suffixes = [ESSAY["tests"][t]["suffix"] for t in tests]
    names = BIRDS + [b + s for s in suffixes for b in BIRDS]
    fps = (fingerprints_gpu if engine == "gpu" else fingerprints_oracle)(names, hop, tail, resampler)
    out = {}
    for t, s in zip(tests, suffixes):
        if engine == "gpu":
            import lbaudiodetective_amd as lb
            fp = {n: lb.Fingerprint.from_bools(v) for n, v in fps.items() if n in BIRDS or n.endswith(s)}
            m = [[fp[a].compare_to_fingerprint(fp[b + s], 200) * 100.0 for b in BIRDS] for a in BIRDS]
        else:
            m = [[O.compare_fp(fps[a], fps[b + s], 200) * 100.0 for b in BIRDS] for a in BIRDS]
        out[t] = np.array(m, np.float64)
    return out


def summarize(ms):
    rows = {}
    for t, m in ms.items():
        e = np.array(ESSAY["tests"][t]["right"])
        d = np.diag(m)
        rows[t] = {"right": [round(float(v), 2) for v in d], "essay": list(e),
                   "abs_dev": [round(float(v), 2) for v in np.abs(d - e)],
                   "identified": int((m.argmax(axis=1) == np.arange(10)).sum()),
                   "essay_identified": ESSAY["tests"][t]["identified"],
                   "wrong_max": round(float(m[~np.eye(10, dtype=bool)].max()), 2),
                   "wrong_min": round(float(m[~np.eye(10, dtype=bool)].min()), 2)}
    return rows


def check(ms):
    """The bounds tests/ assert.  Returns a list of violations (empty = pass)."""
    bad = []
    if "test1" in ms:
        m = ms["test1"]
        d, e = np.diag(m), np.array(ESSAY["tests"]["test1"]["right"])
        if not (m.argmax(axis=1) == np.arange(10)).all():
            bad.append("test1: not every original matches its own _eql sequence best")
        for i, b in enumerate(BIRDS):
            if b in UNREACHABLE_TEST1:
                continue
            if abs(d[i] - e[i]) > 1.0:
                bad.append(f"test1 {b}: {d[i]:.2f} vs essay {e[i]} (more than 1 point)")
        if not 92.0 < d[BIRDS.index("Wren")] <= 100.0 * 20 / 21 + 1e-3:
            bad.append(f"test1 Wren: {d[BIRDS.index('Wren')]:.2f} outside (92, 95.24]")
        if not 50.0 < d[BIRDS.index("Chaffinch")] < 62.0:
            bad.append(f"test1 Chaffinch: {d[BIRDS.index('Chaffinch')]:.2f} outside the re-encoded fixture's band")
        off = m[~np.eye(10, dtype=bool)]
        if not (45.0 < off.min() and off.max() < 58.0):
            bad.append(f"test1: unrelated birds {off.min():.1f}..{off.max():.1f} not at chance level")
    for t, lo, hi in (("test2", 45.0, 58.0), ("test4", 45.0, 58.0)):     # essay: everything 49-54 %
        if t in ms:
            m = ms[t]
            if not (lo < m.min() and m.max() < hi):
                bad.append(f"{t}: matches {m.min():.1f}..{m.max():.1f} outside the chance band")
            d, e = np.diag(m), np.array(ESSAY["tests"][t]["right"])
            if np.abs(d - e).mean() > 2.5:
                bad.append(f"{t}: mean deviation from the essay {np.abs(d - e).mean():.2f} > 2.5 points")
            if (m.argmax(axis=1) == np.arange(10)).sum() > 5:
                bad.append(f"{t}: more than five birds identified (essay: {ESSAY['tests'][t]['identified']})")
    # observed (default models; profiles/r04_birds_sweep.json row header/0/1): 3.06 / 8.21 and 1.84 / 6.60 -> + 1 point
    for t, mae, mx in (("test3_1", 4.1, 9.3), ("test3_2", 2.9, 7.7)):
        if t in ms:
            m = ms[t]
            d, e = np.diag(m), np.array(ESSAY["tests"][t]["right"])
            dev = np.abs(d - e)
            if dev.mean() > mae or dev.max() > mx:
                bad.append(f"{t}: deviation mean {dev.mean():.2f} / max {dev.max():.1f} above {mae} / {mx}")
            # the two birds the essay singles out stay on top, the noise ranking is the essay's
            top2 = set(np.argsort(-d)[:2])
            if top2 != {BIRDS.index("BlackBird"), BIRDS.index("Crow")}:
                bad.append(f"{t}: the two most robust birds are {[BIRDS[i] for i in top2]}, essay: BlackBird, Crow")
            if np.corrcoef(d, e)[0, 1] < 0.8:
                bad.append(f"{t}: correlation with the essay's bars {np.corrcoef(d, e)[0, 1]:.2f} < 0.8")
    return bad


def sweep(path):
    """Every model the essay cannot see directly, through the INDEPENDENT oracle (CPU only): three converters x two IMA4
    packet-start models x three end-of-file treatments; per combination the deviation from the essay's bars and the
    birds identified, test by test."""
    from oracle import oracle as O
    rows = []
    for carry in (0, 1):
        O.lib().lbo_file_set_ima4_carry(carry)
        for resampler in (0, 1, 2):
            for tail in (1, 2, 0):
                ms = matrices("oracle", 1, tail, resampler)
                sm = summarize(ms)
                row = {"ima4_start": "carry" if carry else "header", "resampler": resampler, "tail": tail}
                for t, r in sm.items():
                    dev = np.array(r["abs_dev"])
                    skip = [BIRDS.index(b) for b in UNREACHABLE_TEST1] if t == "test1" else []
                    keep = [i for i in range(10) if i not in skip]
                    row[t] = {"mean_dev": round(float(dev[keep].mean()), 2), "max_dev": round(float(dev[keep].max()), 2),
                              "identified": r["identified"], "essay_identified": r["essay_identified"]}
                rows.append(row)
                print(json.dumps(row), flush=True)
    O.lib().lbo_file_set_ima4_carry(0)
    json.dump({"engine": "oracle (oracle/lbad_file_oracle.c + oracle/lbad_oracle.c)", "hop": 1,
               "note": "test1: the two fixtures that cannot be reached (Chaffinch, Wren) are left out of the deviations",
               "rows": rows}, open(path, "w"), indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--engine", default="gpu", choices=["gpu", "oracle"])
    ap.add_argument("--sweep", help="write the model sweep (oracle engine, CPU) to this JSON file and exit")
    ap.add_argument("--hop", type=int, default=1)
    ap.add_argument("--tail", type=int, default=1)
    ap.add_argument("--resampler", type=int, default=0)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--json")
    ap.add_argument("--matrix", action="store_true", help="print the full 10 x 10 matrices")
    args = ap.parse_args()
    if args.sweep:
        sweep(args.sweep)
        return
    ms = matrices(args.engine, args.hop, args.tail, args.resampler)
    rows = summarize(ms)
    for t in TESTS:
        r = rows[t]
        print(f"{t} ({ESSAY['tests'][t]['suffix']}): identified {r['identified']}/10 (essay {r['essay_identified']}), "
              f"mean |dev| {np.mean(r['abs_dev']):.2f}, max {max(r['abs_dev']):.1f}, wrong birds {r['wrong_min']}..{r['wrong_max']}")
        print("   here :", " ".join(f"{v:5.1f}" for v in r["right"]))
        print("   essay:", " ".join(f"{v:5.1f}" for v in r["essay"]))
        if args.matrix:
            print(np.array2string(ms[t], precision=1, max_line_width=200))
    if args.json:
        json.dump({"engine": args.engine, "hop": args.hop, "tail": args.tail, "resampler": args.resampler, "tests": rows},
                  open(args.json, "w"), indent=1)
    if args.check:
        bad = check(ms)
        for b in bad:
            print("CHECK FAILED:", b)
        print("check:", "FAILED" if bad else "ok")
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
