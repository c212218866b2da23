"""Small ragged corpora against the oracle, entry by entry (bring-up aid for k_sliding.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import lbaudiodetective_amd as lb
from lbaudiodetective_amd import _native as _N
if os.environ.get("LBAD_LIB"):
    _N.LIB_PATH = os.path.abspath(os.environ["LBAD_LIB"])
from oracle import oracle as O

def rand_fp(rng, n, L=200):
    pairs = (L + 1) // 2
    pos = rng.random((n, pairs)) < 0.5
    zero = rng.random((n, pairs)) < 0.03
    f = np.zeros((n, 2 * pairs), np.uint8)
    f[:, 0::2] = pos & ~zero
    f[:, 1::2] = ~pos & ~zero
    return np.ascontiguousarray(f[:, :L])

rng = np.random.default_rng(1)
only = [int(v) for v in os.environ.get("LBAD_DEBUG_NQ", "").split(",") if v]
for lens in ([1, 2, 3], [1] * 5, [5, 9, 13, 30], list(range(1, 40)), [70] * 3 + [20] * 3, list(rng.integers(1, 71, 300))):
    entries = [rand_fp(rng, int(n)) for n in lens]
    counts = np.array(lens, np.uint32)
    c = lb.Corpus.ragged(200, len(entries), int(counts.sum()))
    flat = np.concatenate(entries, axis=0)
    packed = np.stack([lb.pack_subfingerprint(r) for r in flat]).view(np.uint8).reshape(-1, 32)
    c.append_ragged_packed_device(torch.from_numpy(packed).cuda(), counts)
    for nq in (only or (1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 21, 48)):
        q = rand_fp(rng, nq)
        k = min(nq, entries[-1].shape[0])
        q[:k] = entries[-1][:k]
        bi, bs, want = O.corpus_best_ragged(q, entries, 200, want_scores=True)
        got = c.scores_device(lb.Fingerprint.from_bools(q), 0).cpu().numpy()
        bad = np.nonzero(got.view(np.uint32) != want.view(np.uint32))[0]
        tag = "ok " if bad.size == 0 else "BAD"
        print(tag, "lens", lens[:8], "n", len(lens), "nq", nq, "bad", bad[:8].tolist(), "ne", [int(lens[i]) for i in bad[:8]],
              "got", got[bad[:4]].tolist(), "want", want[bad[:4]].tolist(), flush=True)
