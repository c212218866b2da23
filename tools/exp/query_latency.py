"""Latency of the corpus query paths for a given library build: python tools/exp/query_latency.py [lib.so ...]"""
import sys, os, subprocess, json, time
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
import lbaudiodetective_amd._native as N
child = "--child" in sys.argv
libs = [a for a in sys.argv[1:] if a != "--child"] or [N.LIB_PATH]
if not child:
    for l in libs:
        out = subprocess.run([sys.executable, __file__, l, "--child"], capture_output=True, text=True, timeout=300)
        print(os.path.basename(l), out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
    sys.exit(0)
N.LIB_PATH = libs[0]
import torch
import lbaudiodetective_amd as lb
res = {}
for n in (1_000_000, 10_000_000):
    c = lb.Corpus(200, 5, n)
    for b in range(0, n, 1 << 20):
        c.append_packed_device(lb.synth_corpus_device(0x4C424145, b, min(1 << 20, n - b), 5, 200))
    q = lb.Fingerprint.from_bools(lb.unpack_packed(lb.synth_corpus_device(0x4C424145, 777, 1, 5, 200).cpu().numpy(), 200))
    key = torch.zeros(1, dtype=torch.int64, device="cuda")
    for _ in range(5):
        r = c.query(q)
        c.query_key_device(q, key)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        c.query_key_device(q, key)
    e1.record(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        r = c.query(q)
    dt = (time.perf_counter() - t0) / 200
    res[f"n{n//1000000}M"] = {"scan_us": round(e0.elapsed_time(e1) / 20 * 1e3, 1), "query_api_us": round(dt * 1e6, 1), "best": r[0]}
    c.dispose()
print(json.dumps(res))
