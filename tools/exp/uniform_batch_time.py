"""Q x N on a UNIFORM corpus (compare_planes_batch_kernel, k_compare.hip): 1 / 2 / 4 / 8 queries against 10 M entries of 5
sub-fingerprints in one call, HIP events; every key against the single query's."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import lbaudiodetective_amd as lb
SEED = 0x4C424145
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
per = 5
c = lb.Corpus(200, per, n)
chunk = 1 << 20
for b in range(0, n, chunk):
    c.append_packed_device(lb.synth_corpus_device(SEED, b, min(chunk, n - b), per, 200))
fps = []
for k in range(8):
    e = (1_234_567 * (k + 1)) % n
    q = lb.unpack_packed(lb.synth_corpus_device(SEED, e, 1, per, 200).cpu().numpy(), 200)
    fps.append(lb.Fingerprint.from_bools(q.reshape(per, 200)))
single = torch.zeros(8, dtype=torch.int64, device="cuda")
for i, f in enumerate(fps):
    c.query_key_device(f, single[i:i + 1])
torch.cuda.synchronize()
row = {"entries": n}
keys = torch.zeros(8, dtype=torch.int64, device="cuda")
for b in (1, 2, 4, 8):
    for _ in range(3):
        c.query_batch_keys_device(fps[:b], keys)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        c.query_batch_keys_device(fps[:b], keys)
    e1.record()
    torch.cuda.synchronize()
    row[f"batch_of_{b}_ms"] = round(e0.elapsed_time(e1) / 20, 4)
    assert torch.equal(keys[:b], single[:b])
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    c.query_key_device(fps[0], keys[:1])
e1.record()
torch.cuda.synchronize()
row["single_query_ms"] = round(e0.elapsed_time(e1) / 20, 4)
print(json.dumps(row))
