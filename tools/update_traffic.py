#!/usr/bin/env python3
"""profiles/traffic.json from the round's profile summaries (tools/prof_all.sh rNN -> gpurun_out/prof_rNN_*/summary.json, copied
to profiles/rNN_*_summary.json): HBM bytes per clip / record of every kernel the bench line quotes a `traffic` figure for.
    python3 tools/update_traffic.py r06
FETCH_SIZE / WRITE_SIZE are KiB per launch, collected in separate --pmc passes; FETCH_SIZE is doubled (gfx950 counts 64 B per
128-byte request of a 16-B-per-lane streaming read: MI355X_MICROARCH.md, HBM section; calibrated in round 2 on a copy of known
size: 0.50 x, and in round 5 on compare_planes_kernel<5>: 1.978).  Entries of earlier rounds that this round did not re-measure
stay as they are, with their own `round`."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r06"
rnd = int(R[1:])
path = os.path.join(ROOT, "profiles", "traffic.json")
tj = json.load(open(path))
RECORDS = 45002852          # 1 M entries of 20..70 sub-fingerprints (tools/prof_sliding.py)


def summary(label):
    p = os.path.join(ROOT, "profiles", f"{R}_{label}_summary.json")
    return json.load(open(p))["kernels"] if os.path.exists(p) else None


def find(kernels, pattern):
    hits = [(k, v) for k, v in kernels.items() if re.search(pattern, k) and "counters" in v and "FETCH_SIZE" in v["counters"]]
    return max(hits, key=lambda kv: kv[1].get("avg_us", 0) * kv[1].get("calls", 1)) if hits else (None, None)


def per_unit(v, units_per_launch):
    f = v["counters"]["FETCH_SIZE"] * 1024 * 2 / units_per_launch
    w = v["counters"].get("WRITE_SIZE", 0.0) * 1024 / units_per_launch
    return f, w


FLOPS = {"v_pk_fma_f32": 4, "v_pk_mul_f32": 2, "v_pk_add_f32": 2, "v_fma_f32": 2, "v_fmac_f32": 2, "v_mac_f32": 2, "v_add_f32": 1, "v_sub_f32": 1,
         "v_subrev_f32": 1, "v_mul_f32": 1}
_isa = {}


def static_mix(src, pattern):
    """float32 operations per vector instruction of the COMPILED kernel (an FMA = 2, a packed instruction = two lanes' worth):
    the kernels are one loop whose body is almost all of their text, so the static mix stands for the executed one.  hipcc
    cross-compiles here; flags as in lbaudiodetective_amd/csrc/Makefile."""
    import subprocess
    import tempfile
    if src not in _isa:
        out = os.path.join(tempfile.mkdtemp(), "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
               "-fno-slp-vectorize", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-x", "hip", "--cuda-device-only", "-S",
               "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "lbaudiodetective_amd", "csrc", src), "-o", out]
        try:
            subprocess.run(cmd, check=True, capture_output=True, timeout=900)
            _isa[src] = open(out).read()
        except Exception as e:              # noqa: BLE001 -- no compiler: the entry simply carries no mix
            print("static_mix:", src, e)
            _isa[src] = ""
    best = None
    for m in re.finditer(r"^(_ZN4lbad\S+):.*?\n(.*?)s_endpgm", _isa[src], re.M | re.S):
        if not re.search(pattern, m.group(1)):
            continue
        n = f = 0
        for line in m.group(2).splitlines():
            t = line.split()
            if t and t[0].startswith("v_"):
                n += 1
                f += next((w for k, w in FLOPS.items() if t[0].startswith(k)), 0)
        if n and (best is None or n > best[0]):
            best = (n, f)
    return best


def stage(entry, label, pattern, clips_total, algorithmic, note, src=None, mangled=None, windows_per_clip=None):
    ks = summary(label)
    if not ks:
        return
    k, v = find(ks, pattern)
    if not k:
        return
    units = clips_total / v["calls"]
    f, w = per_unit(v, units)
    old = {kk: tj[entry][kk] for kk in ("fetch_bytes_per_clip", "write_bytes_per_clip", "hbm_bytes_per_clip", "round") if entry in tj and kk in tj[entry]}
    tj[entry] = {"kernel": k, "fetch_bytes_per_clip": round(f), "write_bytes_per_clip": round(w), "hbm_bytes_per_clip": round(f + w),
                 "algorithmic_bytes_per_clip": algorithmic, "clips_per_launch": round(units, 1), "kernel_avg_us": v.get("avg_us"),
                 "note": note + f" (profiles/{R}_{label}_summary.json: {v['calls']} launches, FETCH_SIZE {v['counters']['FETCH_SIZE']:.0f} KiB x 2, "
                         f"WRITE_SIZE {v['counters'].get('WRITE_SIZE', 0):.0f} KiB per launch)",
                 "round": rnd, "previous": old or None}
    insts = v["counters"].get("SQ_INSTS_VALU")
    if insts and src:
        mix = static_mix(src, mangled)
        if mix:
            # what the vector ALU ISSUED per clip, in float32 operations (FMA = 2): the counter x 64 lanes x the static mix
            tj[entry].update({"valu_instructions_per_launch": round(insts), "static_vector_instructions": mix[0],
                              "float_ops_per_vector_instruction": round(mix[1] / mix[0], 4),
                              "issued_float_ops_per_clip": round(insts * 64 * mix[1] / mix[0] / units),
                              "issued_float_ops_per_window": round(insts * 64 * mix[1] / mix[0] / units / windows_per_clip) if windows_per_clip else None})


def sliding(entry, label, pattern, note, queries=1):
    ks = summary(label)
    if not ks:
        return
    k, v = find(ks, pattern)
    if not k:
        return
    f, w = per_unit(v, RECORDS)
    old = {kk: tj[entry][kk] for kk in tj.get(entry, {}) if kk in ("fetch_bytes_per_record", "hbm_bytes_per_record", "fetch_bytes_per_record_and_launch", "round", "kernel")}
    e = {"kernel": k, "fetch_bytes_per_record": round(f, 2), "write_bytes_per_record": round(w, 2), "hbm_bytes_per_record": round(f + w, 2),
         "algorithmic_bytes_per_record": 25, "layout_bytes_per_record": 32, "kernel_avg_us": v.get("avg_us"),
         "note": note + f" (profiles/{R}_{label}_summary.json: FETCH_SIZE {v['counters']['FETCH_SIZE']:.0f} KiB x 2 per launch)",
         "round": rnd, "previous": old or None}
    if queries > 1:
        e["queries_per_launch"] = queries
        e["fetch_bytes_per_record_and_query"] = round(f / queries, 2)
    tj[entry] = e


# reps = 3 in tools/prof_all.sh for the stage kernels
stage("stage1_pruned", "B_headline", r"frame_rows_pruned_kernel", 100000 * 3, 176525, "headline: 1 s clips at 44.1 kHz, 1024-sample windows, compact frames out",
      "k_rows_pruned.hip", r"frame_rows_pruned_kernelILi0E", 640)
stage("stage2_select32", "B_headline", r"haar_select32_kernel", 100000 * 3, 176525, "stage 2 of the headline (sparse form, compact frames in)")
stage("stage1_stream_2048", "A_stream2", r"rows_stream2_kernel", 20000 * 3, 198557, "configs[0] settings: 9 s clips at 5512 Hz, 2048-sample windows",
      "k_rows_stream2.hip", r"rows_stream2_kernelILi0E", 640)
stage("stage1_full_2048", "A_rows_full", r"rows_full_kernel", 20000 * 3, 198557, "the same batch through rows_full_kernel (kernel variant 3)")
stage("stage1_stream_4096", "C_stream", r"rows_stream_kernel", 10000 * 3, 192125, "configs[4]: 1 s clips at 48 kHz stereo-summed, 4096-sample windows",
      "k_rows_stream.hip", r"rows_stream_kernelILi0E", 640)
sliding("sliding_q21", "sliding_q21", r"compare_sliding_kernel", "1 M entries of 20..70 sub-fingerprints = 45002852 records, query of 21")
sliding("sliding_q48", "sliding_q48", r"compare_sliding_kernel", "same corpus, query of 48")
sliding("sliding_q5", "sliding_q5", r"compare_short_kernel", "same corpus, query of 5 (systolic scan, one record per lane)")
sliding("sliding_batch8_q21", "sliding_batch8_q21", r"compare_sliding_kernel", "eight queries of 21 = two launches of four queries", queries=4)
sliding("sliding_batch8_q5", "sliding_batch8_q5", r"compare_short_multi_kernel", "eight queries of 5 in ONE launch of compare_short_multi_kernel (round 6)", queries=8)
tj["round"] = rnd
tj[f"source_{R}"] = f"tools/prof_all.sh {R} on MI355X, tools/update_traffic.py {R} (counters are KiB, FETCH_SIZE doubled; entries carry the round they were measured in)"
json.dump(tj, open(path, "w"), indent=1)
print("updated", path)
for k, v in tj.items():
    if isinstance(v, dict) and v.get("round") == rnd:
        print(" ", k, {kk: vv for kk, vv in v.items() if kk.startswith(("fetch", "write", "hbm"))})
