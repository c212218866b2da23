#!/usr/bin/env python3
"""Upstream's own test (LBAudioDetectiveTests.m:53-92) on the bundled bird fixtures: every original
recording against the ten sequences of one suffix through LBAudioDetectiveCompareAudioURLs; prints the
best match per original and the full matrix.    python tools/birds_matrix.py [hop_mode]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lbaudiodetective_amd as lb

BIRDS = ["BlackBird", "BlueTit", "Chaffinch", "Sparrow", "GreatTit", "Crow", "Wren", "Chiffchaff", "Kestrel", "Pigeon"]
D = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "birds")
hop = int(sys.argv[1]) if len(sys.argv) > 1 else 0
det = lb.Detective()
det.set_file_hop_mode(hop)
for suffix in ("_eql", "_dif", "_blu1", "_blu2"):
    m = np.zeros((10, 10), np.float32)
    for i, a in enumerate(BIRDS):
        for j, b in enumerate(BIRDS):
            m[i, j] = det.compare_audio_urls(os.path.join(D, a + ".caf"), os.path.join(D, b + suffix + ".caf"))
    best = m.argmax(axis=1)
    ok = int((best == np.arange(10)).sum())
    off = m[~np.eye(10, dtype=bool)]
    print(f"hop mode {hop} suffix {suffix}: {ok}/10 originals match their own sequence best; "
          f"true matches {np.diag(m).min():.3f}..{np.diag(m).max():.3f}, others {off.min():.3f}..{off.max():.3f}")
    print(np.array2string(m, precision=3, max_line_width=200))
