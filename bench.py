#!/usr/bin/env python3
"""Headline benchmark: audio-seconds fingerprinted per second (BASELINE.json metric).

One "step" = one pass of the fingerprint hot path (frame -> FFT -> sub-band energy -> Haar ->
ranked sign bits) over the whole resident batch: 100 000 synthetic 1 s / 44.1 kHz mono clips,
1024-point windows, stride 64 (BASELINE.json configs[1]).  Inputs are generated on the device
before the timed region, outputs (5 x 32 bytes per clip) stay in HBM.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

With N > 1 every rank fingerprints its own 100 000 clips (weak scaling, no data-path
collective: clips are independent); the timed region is bracketed by barrier + synchronize and
the slowest rank's time is used.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0x4C424144
CSEED = 0x4C424145
RATE, WINDOW, STRIDE, SAMPLES = 44100, 1024, 64, 44100
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP32_PEAK_TFLOPS = 157.3       # vector FP32 spec


def usable_cores() -> int:
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def algorithmic_bytes_per_clip(n_samples: int, window: int, stride: int) -> int:
    """4 L input + 25 bytes of information per sub-fingerprint."""
    per = ((n_samples - window) // stride) // 128
    return 4 * n_samples + 25 * per


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--clips", type=int, default=100_000, help="clips resident per GPU")
    ap.add_argument("--variant", type=int, default=0, help="0 auto, 1 unfused kernels, 2 fused kernel")
    ap.add_argument("--corpus", type=int, default=1_000_000, help="entries for the compare-leg side measurement (0 = skip)")
    ap.add_argument("--corpus-hbm", type=int, default=10_000_000,
                    help="entries per GPU of the HBM-resident scan (larger than the 256 MiB Infinity Cache; 0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="clips for the CPU baseline (0 = 250 per thread)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import lbaudiodetective_amd as lb

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    n_gpus = max(world, 1)
    dev = torch.device("cuda", torch.cuda.current_device())

    det = lb.Detective().configure(sample_rate=RATE, window=WINDOW, stride=STRIDE)
    det.set_kernel_variant(args.variant)
    per = det.subfingerprint_count(SAMPLES)
    n_clips = args.clips

    # ---- resident synthetic input (untimed) --------------------------------------------------
    clips = torch.empty((n_clips, SAMPLES), dtype=torch.float32, device=dev)
    lb.synth_clips_device(SEED, rank * n_clips, n_clips, RATE, SAMPLES, out=clips)
    packed = torch.empty((n_clips, per, lb.PACKED_BYTES), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    def step():
        det.fingerprint_clips_device(clips, out=packed)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    det.set_stage_timing(True)      # HIP events around each kernel, on the launch stream, inside the timed region
    t0 = time.perf_counter()
    for s in range(args.steps):
        ev[s][0].record()
        step()
        ev[s][1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1e3 / args.steps
    value = n_gpus * n_clips * args.steps / elapsed            # audio-seconds per second, whole job
    kernel_ms = sorted(a.elapsed_time(b) for a, b in ev)
    kern_avg_ms = sum(kernel_ms) / len(kernel_ms)
    stage1_ms, stage2_ms, launches = det.stage_times()      # summed over the K timed steps
    det.set_stage_timing(False)
    launches_per_step = launches // args.steps
    rows_ms = stage1_ms / launches                          # dominant kernel: average launch duration
    clips_per_launch = n_clips / launches_per_step

    # ---- parity of the bench's own data against the oracle (untimed, rank 0) -------------------
    result = None
    if rank == 0:
        alg_bytes = algorithmic_bytes_per_clip(SAMPLES, WINDOW, STRIDE)
        achieved = alg_bytes * n_clips / (kern_avg_ms * 1e-3) / 1e9
        canon_flops = per * 128 * 2.5 * WINDOW * 10            # 2.5 W log2 W per window
        # HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE/WRITE_SIZE, corrected as
        # MI355X_MICROARCH.md prescribes); they cannot be collected from inside this process
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = "stage1_pruned" if args.variant != 1 else "stage1_generic"
                if key in tj:
                    traffic = round(tj[key]["hbm_bytes_per_clip"] * clips_per_launch)
            except (OSError, ValueError, KeyError):
                traffic = None
        result = {
            "metric": "audio_seconds_fingerprinted_per_sec",
            "value": round(value, 1),
            "unit": "audio-s/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "configs[1]: 100k synthetic 1 s @44.1 kHz mono clips, 1024-pt FFT, stride 64, "
                            "fingerprint-only, input resident in HBM",
                "clips_per_gpu": n_clips, "samples_per_clip": SAMPLES, "window": WINDOW, "stride": STRIDE,
                "bands": 32, "subfingerprints_per_clip": per, "kernel_variant": args.variant,
                "parallelism": f"clips sharded x{n_gpus}, no collective",
            },
            "per_gpu_value": round(value / n_gpus, 1),
            "roofline": {
                "bound": "hbm",
                "kernel": "stage 1, windows -> frame rows (frame_rows_pruned_kernel when the pruned FFT applies, "
                          "else fft_bands_kernel): dominant kernel of the pass",
                "achieved": round(alg_bytes * clips_per_launch / (rows_ms * 1e-3) / 1e9, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(alg_bytes * clips_per_launch / (rows_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "traffic": traffic,
                "algorithmic_bytes_per_clip": alg_bytes,
                "clips_per_launch": clips_per_launch,
                "launches_per_step": launches_per_step,
                "kernel_ms_avg": round(rows_ms, 4),
                "stage2_kernel_ms_avg": round(stage2_ms / launches, 4),
                "step_ms_avg": round(kern_avg_ms, 4),
                # the pass is FP32-VALU/LDS bound, not HBM bound: canonical-FFT rate beside it
                "fp32_canonical_tflops": round(canon_flops * clips_per_launch / (rows_ms * 1e-3) / 1e12, 3),
                "fp32_peak_tflops": FP32_PEAK_TFLOPS,
                "whole_pass_achieved_GBps": round(achieved, 2),
            },
        }

        from oracle import oracle as O
        cfg = O.Config(RATE, WINDOW, STRIDE)
        n_par = min(64, n_clips)
        host = clips[:n_par].cpu().numpy()
        want = O.fingerprint_batch(host, cfg, nthreads=min(8, usable_cores()))
        got = lb.unpack_packed(packed[:n_par].cpu().numpy(), 200).reshape(n_par, per, 200)
        result["parity"] = {"clips_checked": n_par, "bit_exact": bool(np.array_equal(got, want))}

        if not args.no_cpu_baseline:
            threads = usable_cores()
            n_cpu = args.cpu_sample or min(n_clips, 4000 * threads)     # ~11 s at ~2.8 ms per clip per thread
            sample = clips[:n_cpu].cpu().numpy()
            O.fingerprint_batch(sample[: 2 * threads], cfg, nthreads=threads)    # warm the caches/threads
            t1 = time.perf_counter()
            O.fingerprint_batch(sample, cfg, nthreads=threads)
            dt = time.perf_counter() - t1
            result["cpu_baseline"] = {
                "value": round(n_cpu / dt, 2),
                "unit": "audio-s/s",
                "cores": threads,
                "kind": "port",
                "sample": f"first {n_cpu} clips of the same batch through oracle/lbad_oracle.c "
                          f"(scalar radix-2 restatement, not vDSP), {threads} OpenMP threads, {dt:.1f} s",
            }

    # ---- compare leg (side measurement, BASELINE configs[2]/[3]) -------------------------------
    if args.corpus > 0:
        n_local = args.corpus
        sc = lb.ShardedCorpus(200, per, n_local * n_gpus, rank=rank, world_size=n_gpus)
        chunk = 1 << 20
        for b in range(sc.begin, sc.end, chunk):
            m = min(chunk, sc.end - b)
            sc.append_packed_device(lb.synth_corpus_device(CSEED, b, m, per, 200))
        planted = (777_777 % (n_local * n_gpus))
        qsrc = lb.unpack_packed(lb.synth_corpus_device(CSEED, planted, 1, per, 200).cpu().numpy(), 200)
        rng = np.random.default_rng(7)
        flip = rng.random((per, 100)) < 0.07
        q = qsrc.copy()
        q[:, 0::2] = np.where(flip, qsrc[:, 1::2], qsrc[:, 0::2])
        q[:, 1::2] = np.where(flip, qsrc[:, 0::2], qsrc[:, 1::2])
        fq = lb.Fingerprint.from_bools(q)
        key = torch.zeros(1, dtype=torch.int64, device=dev)
        for _ in range(3):
            best = sc.query(fq, key_out=key)
        barrier()
        reps = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t1 = time.perf_counter()
        e0.record()
        for _ in range(reps):
            sc.local.query_key_device(fq, key, 0, index_base=sc.begin)
        e1.record()
        torch.cuda.synchronize()
        scan_ms = e0.elapsed_time(e1) / reps
        t1 = time.perf_counter()
        for _ in range(reps):
            best = sc.query(fq, key_out=key)
        lat_ms = (time.perf_counter() - t1) * 1e3 / reps
        if rank == 0:
            result["compare"] = {
                "workload": f"1 query vs {n_local * n_gpus} fingerprints ({per} x 200 Booleans), sharded x{n_gpus}",
                "best_index": best[0], "best_score": best[1], "planted_index": planted,
                "scan_ms": round(scan_ms, 4),
                "query_latency_ms": round(lat_ms, 4),
                "entries_per_s": round(n_local * n_gpus / (scan_ms * 1e-3), 1),
                "achieved_GBps_algorithmic": round(25 * per * n_local / (scan_ms * 1e-3) / 1e9, 2),
                "achieved_GBps_layout": round(sc.local.entry_stride_bytes * n_local / (scan_ms * 1e-3) / 1e9, 2),
                "hbm_frac_algorithmic": round(25 * per * n_local / (scan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            }
            if world == 1 and not args.no_cpu_baseline:
                # the same query over the same corpus through the oracle's Boolean-per-byte loop
                # (the reference's layout: 1000 B per entry), all usable cores; also a full-size parity check
                from oracle import oracle as O
                threads = usable_cores()
                n_cmp = n_local
                host = np.empty((n_cmp, per, 200), np.uint8)
                for b in range(0, n_cmp, 1 << 18):
                    m = min(1 << 18, n_cmp - b)
                    host[b:b + m] = lb.unpack_packed(
                        lb.synth_corpus_device(CSEED, b, m, per, 200).cpu().numpy(), 200).reshape(m, per, 200)
                O.corpus_best(q, host[: 1 << 14], 200, nthreads=threads)
                t1 = time.perf_counter()
                ci, cs = O.corpus_best(q, host, 200, nthreads=threads)
                dt = time.perf_counter() - t1
                del host
                result["compare"]["cpu_baseline"] = {
                    "value": round(n_cmp / dt, 1), "unit": "entries/s", "cores": threads, "kind": "port",
                    "sample": f"all {n_cmp} entries as {per} x 200 Booleans (the reference's layout) through "
                              f"oracle/lbad_oracle.c:lbo_corpus_best, {threads} OpenMP threads, {dt * 1e3:.0f} ms",
                }
                result["compare"]["parity"] = {
                    "entries_checked": n_cmp,
                    "bit_exact": bool(ci == best[0] and np.float32(cs).view(np.uint32) == np.float32(best[1]).view(np.uint32)),
                }

    # HBM-resident scan: the 1 M corpus (128 MB) fits the 256 MiB Infinity Cache, this one does not
    if args.corpus > 0 and args.corpus_hbm > 0 and world == 1:
        n_big = args.corpus_hbm
        big = lb.Corpus(200, per, n_big)
        for b in range(0, n_big, 1 << 20):
            big.append_packed_device(lb.synth_corpus_device(CSEED, b, min(1 << 20, n_big - b), per, 200))
        key = torch.zeros(1, dtype=torch.int64, device=dev)
        for _ in range(3):
            big.query_key_device(fq, key)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            big.query_key_device(fq, key)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        idx, sc = lb.Corpus.decode_key(int(key.item()))
        result["compare_hbm"] = {
            "workload": f"1 query vs {n_big} fingerprints on one GPU ({big.entry_stride_bytes * n_big / 1e9:.2f} GB, HBM-resident)",
            "scan_ms": round(ms, 4), "best_index": idx, "best_score": sc,
            "roofline": {"bound": "hbm", "achieved": round(25 * per * n_big / (ms * 1e-3) / 1e9, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(25 * per * n_big / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "layout_GBps": round(big.entry_stride_bytes * n_big / (ms * 1e-3) / 1e9, 1)},
        }
        big.dispose()

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
