#!/usr/bin/env python3
"""Headline benchmark: audio-seconds fingerprinted per second (BASELINE.json metric).

One pass = the fingerprint hot path (frame -> FFT -> sub-band energy -> Haar -> ranked sign bits) over the
whole resident batch: 100 000 synthetic 1 s / 44.1 kHz mono clips, 1024-point windows, stride 64
(BASELINE.json configs[1]).  Inputs are generated on the device before the timed region, outputs (5 x 32 bytes
per clip) stay in HBM.  One "step" = `passes_per_step` back-to-back passes over that batch: a pass takes ~20 ms,
so a step is sized (from two calibration passes, before the warm-up) such that the K timed steps last at least
--min-seconds (default 3 s) -- long enough for clocks and power to settle and for an outside sampler to see the
GPU busy.  `value` is audio-seconds per second over the whole timed region either way; `ms_per_pass`,
`passes_per_step`, the first and last quartile of the step times, the shader clock read inside a probe kernel
and the board power sampled during the timed region are reported beside it.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher: this process -- before it touches torch.cuda or HIP -- starts N fresh ranks
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py ...`),
relays rank 0's single JSON line and exits non-zero if any rank failed.  Under an external launcher
(WORLD_SIZE set) it is one of the ranks; WORLD_SIZE must equal --gpus.

With N > 1 every rank fingerprints its own 100 000 clips (weak scaling, no data-path collective: clips are
independent); the timed region is bracketed by barrier + synchronize and the slowest rank's time is used.
The compare leg (a side measurement) is BASELINE configs[2] at N = 1 (1 query vs 1 M fingerprints) and
configs[3] at N > 1: 10 M fingerprints in contiguous index shards of 10 M / N per rank, the planted match in
a non-zero rank, one RCCL MAX all-reduce of the 8-byte (score, ~index) key, timed separately from the scan.

`--backend gloo --clips 0` is the CPU dry run of the launcher, the rendezvous, the shard arithmetic and the
key reduction (no kernels run, `value` is null); tests/test_bench_launcher.py drives it at world size 2.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import struct
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0x4C424144
CSEED = 0x4C424145
RATE, WINDOW, STRIDE, SAMPLES = 44100, 1024, 64, 44100
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP32_PEAK_TFLOPS = 157.3       # vector FP32 spec
PLANTED_1GPU = 777_777
PLANTED_SHARDED = 7_777_777    # config 4: lies in rank N-1..1 for N = 2, 4, 8 (never rank 0)


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


# Float operations a stage-1 kernel EXECUTES per window beside SURVEY 8(d)'s canonical count (2.5 W log2 W of a real radix-2
# FFT, what the metric is priced in).  Until round 5 this was a constant typed from the kernels' header comments; the round-5
# review asked for the counter.  Now: profiles/traffic.json carries, per stage-1 kernel, SQ_INSTS_VALU per launch (committed
# rocprofv3 --pmc pass) x 64 lanes x the float operations per vector instruction of the COMPILED kernel (FMA = 2, packed = 2
# lanes; tools/update_traffic.py) = the float operations the vector ALU ISSUED per window.  (The typed constants were 10.3 k,
# 55.6 k and 73.6 k operations per window; the counters say 15.8 k, 44.6 k and 91.6 k -- off by 0.65 .. 1.5 x, because a
# butterfly that is "10 operations" on paper is four packed FMAs = 16 issued ones, and shared stages were counted twice.)
STAGE1_TRAFFIC_KEY = {(44100, 1024): "stage1_pruned", (5512, 2048): "stage1_stream_2048", (48000, 4096): "stage1_stream_4096"}


def algorithmic_bytes_per_clip(n_samples: int, window: int, stride: int) -> int:
    """4 L input + 25 bytes of information per sub-fingerprint."""
    per = ((n_samples - window) // stride) // 128
    return 4 * n_samples + 25 * per


def executed_fields(rate: int, window: int, canonical_tflops: float, variant: int) -> dict:
    """fp32_executed_tflops / fp32_frac_executed beside the canonical figures (the specialised kernels only): the float
    operations the vector ALU issued (FMA = 2), from the committed SQ_INSTS_VALU pass and the compiled kernel's instruction mix."""
    none = {"fp32_executed_tflops": None, "fp32_frac_executed": None}
    key = STAGE1_TRAFFIC_KEY.get((rate, window)) if variant != 1 else None
    if key is None:
        return none
    try:
        raw = open(os.path.join(ROOT, "profiles", "traffic.json"), "rb").read()
        e = json.loads(raw)[key]
        issued = float(e["issued_float_ops_per_window"])
    except (OSError, ValueError, KeyError, TypeError):
        return none
    f = issued / (2.5 * window * (window.bit_length() - 1))
    return {"fp32_executed_tflops": round(canonical_tflops * f, 3), "fp32_frac_executed": round(canonical_tflops * f / FP32_PEAK_TFLOPS, 4),
            "executed_over_canonical": round(f, 3),
            "executed_count_source": {"file": "profiles/traffic.json", "sha256": hashlib.sha256(raw).hexdigest()[:16], "entry": key, "round": e.get("round"),
                                      "issued_float_ops_per_window": issued, "valu_instructions_per_launch": e.get("valu_instructions_per_launch"),
                                      "float_ops_per_vector_instruction": e.get("float_ops_per_vector_instruction"),
                                      "how": "SQ_INSTS_VALU (committed rocprofv3 --pmc pass) x 64 lanes x float operations per vector "
                                             "instruction of the compiled kernel (FMA = 2) / windows per launch"}}


def per_call_stage_times(stage1_ms_sum: float, stage2_ms_sum: float, launches: int, calls: int):
    """LBAudioDetectiveGetStageTimes sums the two kernels' durations over every LAUNCH since timing was switched on,
    and a call runs as several launches (chunks) when the inter-stage buffer limit cuts the batch.  Returns what
    ONE call costs -- (stage-1 ms, stage-2 ms, launches per call) -- and the average duration of one launch of each
    kernel: (stage-1 ms per launch, stage-2 ms per launch)."""
    if calls <= 0 or launches <= 0 or launches % calls:
        raise ValueError(f"{launches} launches do not divide into {calls} calls")
    return (stage1_ms_sum / calls, stage2_ms_sum / calls, launches // calls), (stage1_ms_sum / launches, stage2_ms_sum / launches)


def self_check(result: dict) -> list:
    """Every figure of the line that has a physical ceiling, checked against it: a fraction above 1, a TFLOP/s
    figure above the FP32 vector peak, a GB/s figure above the HBM peak or a roofline whose `achieved` exceeds its
    `peak` means the arithmetic behind the line is wrong (round 3: a per-launch time multiplied by a whole batch).
    Returns the list of violations (empty = the line may be printed as it is)."""
    bad = []

    def walk(node, path):
        if isinstance(node, dict):
            if {"achieved", "peak"} <= node.keys() and isinstance(node["achieved"], (int, float)) \
                    and isinstance(node["peak"], (int, float)) and node["achieved"] > node["peak"]:
                bad.append(f"{path}: achieved {node['achieved']} > peak {node['peak']}")
            for k, v in node.items():
                walk(v, f"{path}.{k}" if path else k)
        elif isinstance(node, list):
            for i, v in enumerate(node):
                walk(v, f"{path}[{i}]")
        elif isinstance(node, (int, float)) and not isinstance(node, bool):
            key = path.rsplit(".", 1)[-1].lower()
            if "frac" in key and node > 1.0:
                bad.append(f"{path} = {node} > 1")
            if "tflops" in key and "peak" not in key and node > FP32_PEAK_TFLOPS:
                bad.append(f"{path} = {node} > FP32 vector peak {FP32_PEAK_TFLOPS}")
            if "gbps" in key and node > HBM_PEAK_GBS:
                bad.append(f"{path} = {node} > HBM peak {HBM_PEAK_GBS}")
    walk(result, "")
    return bad


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--clips", type=int, default=100_000, help="clips resident per GPU (0 = skip the fingerprint leg)")
    ap.add_argument("--variant", type=int, default=0, help="0 auto, 1 generic kernels, 2 specialised kernels")
    ap.add_argument("--corpus", type=int, default=-1,
                    help="entries of the compare leg, WHOLE job (-1 = 1 M at one GPU, 10 M sharded; 0 = skip)")
    ap.add_argument("--corpus-hbm", type=int, default=10_000_000,
                    help="one-GPU runs: entries of the extra HBM-resident scan (larger than the 256 MiB Infinity Cache; 0 = skip)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = CPU dry run of launcher + key reduction (needs --clips 0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the side measurements of the configs[0] settings and of configs[4]")
    ap.add_argument("--cpu-sample", type=int, default=0, help="clips for the CPU baseline (0 = 4000 per thread)")
    ap.add_argument("--min-seconds", type=float, default=3.0,
                    help="lower bound of the timed region; a step becomes as many passes over the batch as that takes")
    ap.add_argument("--passes-per-step", type=int, default=0, help="override the calibration (0 = from --min-seconds)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (and take every collective branch) even at one rank: executes the "
                         "N > 1 plumbing -- init_process_group, all_gather_object, the MAX all-reduces, the id broadcast of "
                         "the library's communicator, barrier, destroy -- on a single GPU")
    ap.add_argument("--no-sliding", action="store_true", help="skip the ragged-corpus (sliding compare) leg")
    ap.add_argument("--no-files", action="store_true", help="skip the file leg (BASELINE configs[0] on the bird fixtures)")
    return ap.parse_args(argv)


def quiet_stdout(fn):
    """Run fn with file descriptor 1 pointing at stderr (C libraries that print to stdout must not disturb the
    one JSON line this program owes its caller)."""
    import ctypes
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        return fn()
    finally:
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:                                   # noqa: BLE001
            pass
        os.dup2(saved, 1)
        os.close(saved)


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


class Telemetry:
    """Board power, shader clock and busy percentage of THIS GPU (amdgpu's sysfs files under its PCI address;
    rocm-smi as fallback) and the shader clock read inside a one-wave probe kernel on a side stream, sampled
    from a thread while the timed region runs."""

    def __init__(self, lb, torch, period=0.05):
        import threading
        self.lb, self.torch, self.period = lb, torch, period
        self.stop_flag = False
        self.smi, self.mhz = [], []
        self.side = torch.cuda.Stream()
        self.dev = torch.cuda.current_device()
        self.sysfs = self.find_sysfs(torch, self.dev)
        self.thread = threading.Thread(target=self._run, daemon=True)
        # the probe kernel waits for a free wave slot next to the running kernel (its stream synchronisation can
        # take as long as a pass): its own thread, so that the sysfs samples keep their period
        self.probe_thread = threading.Thread(target=self._probe, daemon=True)

    @staticmethod
    def find_sysfs(torch, dev):
        import glob
        try:
            pr = torch.cuda.get_device_properties(dev)
            addr = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            base = os.path.join("/sys/bus/pci/devices", addr)
            if not os.path.isdir(base):
                return None
            hw = glob.glob(os.path.join(base, "hwmon", "hwmon*"))
            files = {"busy": os.path.join(base, "gpu_busy_percent")}
            for h in hw:
                for name in ("power1_average", "power1_input"):
                    if os.path.exists(os.path.join(h, name)) and "power" not in files:
                        files["power"] = os.path.join(h, name)
                if os.path.exists(os.path.join(h, "freq1_input")):
                    files["sclk"] = os.path.join(h, "freq1_input")
            return {"pci": addr, **{k: v for k, v in files.items() if os.path.exists(v)}}
        except Exception:                                   # noqa: BLE001
            return None

    def read_smi(self):
        out = {}
        if self.sysfs:
            for key, scale, name in (("power", 1e-6, "power_w"), ("sclk", 1e-6, "sclk_mhz"), ("busy", 1.0, "busy_pct")):
                try:
                    out[name] = int(open(self.sysfs[key]).read()) * scale
                except (KeyError, OSError, ValueError):
                    pass
            if out:
                return out
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5)
            card = next(iter(json.loads(r.stdout).values()))
            for k, v in card.items():
                kl = k.lower()
                if "power" in kl and "(w)" in kl:
                    out["power_w"] = float(v)
                elif kl.startswith("sclk clock speed"):
                    out["sclk_mhz"] = float(str(v).strip("()").lower().replace("mhz", ""))
        except Exception:                                   # noqa: BLE001
            pass
        return out

    def _run(self):
        self.torch.cuda.set_device(self.dev)
        while not self.stop_flag:
            t = time.perf_counter()
            v = self.read_smi()
            if v:
                self.smi.append(v)
            time.sleep(max(0.0, self.period - (time.perf_counter() - t)))

    def _probe(self):
        self.torch.cuda.set_device(self.dev)
        while not self.stop_flag:
            try:
                self.mhz.append(self.lb.probe_shader_clock(self.side, 2000))
            except Exception:                               # noqa: BLE001
                pass
            time.sleep(0.25)

    def start(self):
        self.idle = self.read_smi()
        self.thread.start()
        self.probe_thread.start()

    def stop(self):
        self.stop_flag = True
        self.thread.join(timeout=10)
        self.probe_thread.join(timeout=10)

    def summary(self):
        def stats(xs):
            xs = sorted(xs)
            return None if not xs else {"min": round(xs[0], 1), "median": round(xs[len(xs) // 2], 1), "max": round(xs[-1], 1),
                                        "samples": len(xs)}
        return {"shader_clock_mhz_in_kernel": stats(self.mhz),
                "board_power_w": stats([v["power_w"] for v in self.smi if "power_w" in v]),
                "sclk_mhz_driver": stats([v["sclk_mhz"] for v in self.smi if "sclk_mhz" in v]),
                "gpu_busy_pct": stats([v["busy_pct"] for v in self.smi if "busy_pct" in v]),
                "just_before_the_timed_region": self.idle, "source": (self.sysfs or {}).get("pci", "rocm-smi"),
                "how": "probe kernel (s_memtime against the 100 MHz s_memrealtime, 2 ms, side stream) and amdgpu's hwmon / "
                       "gpu_busy_percent files of this GPU's PCI address, sampled from a thread during the timed region"}


# ---- launcher ---------------------------------------------------------------------------------------
def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args) -> int:
    """Parent of an N-rank run.  Nothing here imports torch or touches the GPU: the ranks are fresh child
    processes of `torch.distributed.run`, which this process only waits for."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:
        s = out.strip()
        if s.startswith("{") and '"metric"' in s:
            line = s                        # rank 0's result; everything else is passed through on stderr
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if rc != 0:
        sys.stderr.write(f"bench.py: torch.distributed.run exited with {rc}\n")
        if line is not None:                # a rank refused its own line (self-check, collective fallback): show it, fail
            print(line, flush=True)
        return rc
    if line is None:
        sys.stderr.write("bench.py: the ranks finished without a result line\n")
        return 1
    result = json.loads(line)
    if result.get("n_gpus") != args.gpus or result.get("rccl_ranks") != args.gpus:
        sys.stderr.write(f"bench.py: asked for {args.gpus} ranks, result reports {result.get('rccl_ranks')}\n")
        return 1
    print(line, flush=True)
    return 0


def profile_traffic(variant: int, clips_per_launch: float, key: str = None):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in separate
    passes, corrected as MI355X_MICROARCH.md prescribes).  They cannot be collected from inside this
    process, so the value is only reported together with the file it comes from and that file's hash.
    key: the entry of profiles/traffic.json (default: the headline's stage-1 kernel)."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        raw = open(tpath, "rb").read()
        tj = json.loads(raw)
        if key is None:
            key = "stage1_pruned" if variant != 1 else "stage1_generic"
        if key not in tj:
            return None, None
        return (round(tj[key]["hbm_bytes_per_clip"] * clips_per_launch),
                {"file": "profiles/traffic.json", "sha256": hashlib.sha256(raw).hexdigest()[:16], "entry": key,
                 "round": tj[key].get("round", tj.get("round")), "measured_by": "committed rocprofv3 --pmc passes, not this run"})
    except (OSError, ValueError, KeyError):
        return None, None



# ---- side legs (one GPU) --------------------------------------------------------------------------------
def sliding_leg(args, torch, np):
    """Upstream's best-match loop in its real shape (LBAudioDetectiveTests.m:57-91: one fingerprint against
    candidates of OTHER lengths, LBAudioDetectiveFingerprint.m:123-146 swaps and slides) as one corpus query:
    1 M synthetic entries of 20..70 sub-fingerprints, query of 21 cut out of entry 777 777 with 7 % of its sign
    pairs flipped."""
    import lbaudiodetective_amd as lb
    from oracle import oracle as O
    n, lo, hi, nq = 1_000_000, 20, 70, 21
    counts = O.synth_ragged_counts(CSEED, 0, n, lo, hi)
    total = int(counts.sum())
    packed = lb.synth_ragged_corpus_device(CSEED, 0, counts, 200)
    corpus = lb.Corpus.ragged(200, n, total)
    corpus.append_ragged_packed_device(packed, counts)
    del packed
    src = O.synth_entry(CSEED, PLANTED_1GPU, int(counts[PLANTED_1GPU]), 200)
    q = src[3:3 + nq].copy()
    flip = np.random.default_rng(3).random((nq, 100)) < 0.07
    pos = q[:, 0::2].copy()
    q[:, 0::2] = np.where(flip, q[:, 1::2], pos)
    q[:, 1::2] = np.where(flip, pos, q[:, 1::2])
    fq = lb.Fingerprint.from_bools(q)
    key = torch.zeros(1, dtype=torch.int64, device="cuda")

    def timed_scan(fp, reps=50):
        for _ in range(3):
            corpus.query_key_device(fp, key)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            corpus.query_key_device(fp, key)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps, lb.Corpus.decode_key(int(key.item()) & (2**64 - 1))

    # `scan_ms` and the roofline are the FULL scan: every sliding offset of every entry evaluated (bound pruning off).  The
    # library's default for top-1 queries drops groups of offsets that cannot reach the best match published so far (exact;
    # LBAudioDetectiveCorpusSetBoundPruning): reported beside it, with the planted match and with a query that matches nothing.
    corpus.set_bound_pruning(False)
    ms, best = timed_scan(fq)
    corpus.set_bound_pruning(True)
    ms_pruned, best_pruned = timed_scan(fq)
    stranger = lb.Fingerprint.from_bools(O.synth_entry(CSEED ^ 0x5555, 123, nq, 200))
    ms_stranger, best_stranger = timed_scan(stranger)
    corpus.set_bound_pruning(False)
    ms_stranger_full, best_stranger_full = timed_scan(stranger)
    corpus.set_bound_pruning(True)
    t1 = time.perf_counter()
    for _ in range(20):
        api = corpus.query(fq)
    lat_ms = (time.perf_counter() - t1) * 1e3 / 20
    scores = corpus.scores_device(fq).cpu().numpy()
    alg = 25 * total                                     # 25 B per sub-fingerprint
    traffic, traffic_src = None, None
    try:                                                 # HBM bytes from the committed PMC passes, with provenance
        raw = open(os.path.join(ROOT, "profiles", "traffic.json"), "rb").read()
        tj = json.loads(raw)
        traffic = round(tj["sliding_q21"]["hbm_bytes_per_record"] * total)
        traffic_src = {"file": "profiles/traffic.json", "sha256": hashlib.sha256(raw).hexdigest()[:16], "entry": "sliding_q21",
                       "round": tj["sliding_q21"].get("round"), "measured_by": "committed rocprofv3 --pmc passes, not this run"}
    except (OSError, ValueError, KeyError):
        pass
    out = {
        "workload": f"1 query of {nq} sub-fingerprints vs {n} entries of {lo}..{hi} sub-fingerprints "
                    f"({total} records, {32 * total / 1e9:.2f} GB in HBM), every sliding offset of every entry",
        "best_index": best[0], "best_score": best[1], "planted_index": PLANTED_1GPU, "found_planted": bool(best[0] == PLANTED_1GPU),
        "scan_ms": round(ms, 4), "query_latency_ms": round(lat_ms, 4),
        "with_bound_pruning": {"what": "the library's default for top-1 queries: exact, data-dependent; query_latency_ms above runs with it",
                               "scan_ms_planted_match": round(ms_pruned, 4), "same_result": bool(best_pruned == best),
                               "scan_ms_query_that_matches_nothing": round(ms_stranger, 4),
                               "full_scan_ms_query_that_matches_nothing": round(ms_stranger_full, 4),
                               "same_result_no_match": bool(best_stranger == best_stranger_full),
                               "best_score_no_match": best_stranger[1]},
        "subfingerprint_compares_per_s": round(nq * total / (ms * 1e-3), 1),
        "roofline": {"bound": "valu", "kernel": "compare_sliding_kernel<FULL, false, QLDS, 1, 1024> (k_sliding.hip: only the sliding "
                     "offsets that exist; 2 v_bitop3 + 1 v_bcnt per 32 sign pairs, 8 DPP moves per step of 4 pairs; round 5: a pass's "
                     "windows fetched as whole lines through LDS, query in the kernel arguments, no memset node): integer VALU issue, not HBM",
                     "achieved": round(alg / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes": alg, "layout_GBps": round(32 * total / (ms * 1e-3) / 1e9, 1)},
    }
    assert api == best, (api, best)
    # the same corpus against queries of other lengths (scan only): short queries are HBM-bound, a long one meets
    # entries shorter and longer than itself in the same chunk
    others = {}
    for n_other in (5, 48):
        qo = lb.Fingerprint.from_bools(O.synth_entry(CSEED, PLANTED_1GPU, max(int(counts[PLANTED_1GPU]), n_other), 200)[:n_other])
        corpus.set_bound_pruning(False)
        mo, _ = timed_scan(qo, 20)
        corpus.set_bound_pruning(True)
        others[f"query_of_{n_other}"] = {"scan_ms": round(mo, 4), "algorithmic_GBps": round(alg / (mo * 1e-3) / 1e9, 1),
                                         "layout_GBps": round(32 * total / (mo * 1e-3) / 1e9, 1)}
    out["other_query_lengths"] = others
    # Q x N, the shape of the reference's own test (LBAudioDetectiveTests.m:57-91: ten originals against ten candidates):
    # eight queries of one length in ONE call (four per pass of the task scan; queries of up to 12 sub-fingerprints eight per
    # launch of compare_short_multi_kernel, round 6) against eight single calls -- which run through OTHER kernels (the task
    # scan / the systolic scan of one query): every key of the batch must equal the single query's
    corpus.set_bound_pruning(False)
    batches = {}
    for n_b in (21, 5):
        fps = []
        for k in range(8):
            e = 100_000 * (k + 1) + 777
            fps.append(lb.Fingerprint.from_bools(O.synth_entry(CSEED, e, max(int(counts[e]), n_b), 200)[:n_b]))
        keys8, single8 = torch.zeros(8, dtype=torch.int64, device="cuda"), torch.zeros(8, dtype=torch.int64, device="cuda")
        t_single = 0.0
        for i, f in enumerate(fps):
            mo, _ = timed_scan(f, 5)
            t_single += mo
            corpus.query_key_device(f, single8[i:i + 1])
        for _ in range(2):
            corpus.query_batch_keys_device(fps, keys8)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            corpus.query_batch_keys_device(fps, keys8)
        e1.record()
        torch.cuda.synchronize()
        mb = e0.elapsed_time(e1) / 10
        batches[f"eight_queries_of_{n_b}"] = {
            "one_call_ms": round(mb, 4), "eight_single_calls_ms": round(t_single, 4), "speedup": round(t_single / mb, 2),
            "times_one_query": round(mb / (t_single / 8), 2), "same_keys": bool(torch.equal(keys8, single8)),
            # (the records are fetched once per launch of four / eight queries: bytes over time is NOT eight times an HBM rate)
            "launches": 1 if n_b <= 12 else 2,
            "kernel": ("compare_short_multi_kernel<8, %d>" % n_b) if n_b <= 12 else "compare_sliding_kernel<FULL, false, true, 4, 1024> x 2", "subfingerprint_compares_per_s": round(8 * n_b * total / (mb * 1e-3), 1)}
    corpus.set_bound_pruning(True)
    out["query_batches"] = batches
    if not args.no_cpu_baseline:
        # the oracle's Boolean-per-byte loop (the reference's layout) on a bounded sample of the same corpus;
        # also the parity check of the scores the GPU produced for those entries
        threads = usable_cores()
        n_s = 60_000
        ent = O.synth_ragged_entries(CSEED, 0, counts[:n_s], 200)
        O.corpus_best_ragged(q, (ent[: int(counts[:2000].sum())], counts[:2000]), 200, nthreads=threads)
        t1 = time.perf_counter()
        _, _, want = O.corpus_best_ragged(q, (ent, counts[:n_s]), 200, nthreads=threads, want_scores=True)
        dt = time.perf_counter() - t1
        n_1 = 6_000
        t1 = time.perf_counter()
        O.corpus_best_ragged(q, (ent[: int(counts[:n_1].sum())], counts[:n_1]), 200, nthreads=1)
        dt1 = time.perf_counter() - t1
        out["cpu_baseline"] = {
            "value": round(n_s / dt, 1), "unit": "entries/s", "cores": threads, "kind": "port", "per_core": round(n_s / dt / threads, 1),
            "single_thread": {"value": round(n_1 / dt1, 1), "unit": "entries/s", "cores": 1},
            "cpu_model": cpu_model(),
            "sample": f"first {n_s} entries ({int(counts[:n_s].sum())} sub-fingerprints as 200 Booleans each, the "
                      f"reference's layout) through oracle/lbad_oracle.c:lbo_corpus_best_ragged, {threads} OpenMP threads, {dt * 1e3:.0f} ms",
        }
        out["parity"] = {"entries_checked": n_s,
                         "bit_exact": bool(np.array_equal(scores[:n_s].view(np.uint32), want.view(np.uint32)))}
    return out


def files_leg(args, torch, np):
    """BASELINE configs[0] as worded: bundled Birds/*.caf files through LBAudioDetectiveCompareAudioURLs at the
    reference's defaults (5512 Hz, 2048-point windows, upstream's file loop), then every fixture through the file
    entry points for a files-per-second figure (60 files per call, and 6000 files in one call).  The CPU baseline is the
    independent oracle end to end, its own decoder and converter included."""
    import lbaudiodetective_amd as lb
    from oracle import oracle as O
    birds = os.path.join(ROOT, "tests", "golden", "birds")
    a, b = os.path.join(birds, "BlackBird.caf"), os.path.join(birds, "BlackBird_eql.caf")
    if not (os.path.exists(a) and os.path.exists(b)):
        return None
    det = lb.Detective()
    m = det.compare_audio_urls(a, b)
    reps = 50
    t1 = time.perf_counter()
    for _ in range(reps):
        m = det.compare_audio_urls(a, b)
    pair_ms = (time.perf_counter() - t1) * 1e3 / reps
    paths = sorted(os.path.join(birds, f) for f in os.listdir(birds) if f.endswith(".caf"))
    seconds = 0.0
    for p in paths:
        x, rate = lb.read_audio_url(p)
        seconds += x.size / rate
    import ctypes as C
    from lbaudiodetective_amd import _native as N
    L = N.lib()

    def c_call(batch, rounds_, warm):
        """LBAudioDetectiveProcessAudioURLs itself: the path array is built once, the fingerprints are released outside the
        clock (what a C host sees; the ctypes mirror adds ~8 us of Python per file on top)."""
        n_ = len(batch)
        arr = (C.c_char_p * n_)(*[p.encode() for p in batch])
        tot = 0.0
        for r in range(warm + rounds_):
            refs, sts = (N.Ref * n_)(), (N.OSStatus * n_)()
            t1 = time.perf_counter()
            rc = L.LBAudioDetectiveProcessAudioURLs(det._ref, arr, n_, refs, sts)
            dt_ = time.perf_counter() - t1
            if rc != 0 or any(int(x) != 0 for x in sts):
                raise RuntimeError("LBAudioDetectiveProcessAudioURLs failed inside the bench")
            for i in range(n_):
                L.LBAudioDetectiveFingerprintDispose(refs[i])
            if r >= warm:
                tot += dt_
        return tot / rounds_

    rounds = 20
    fps = det.process_audio_urls(paths)                       # (through the mirror once: the objects the parity check reads)
    dt60 = c_call(paths, rounds, 3)
    t1 = time.perf_counter()
    for _ in range(5):
        det.process_audio_urls(paths)
    dt60_py = (time.perf_counter() - t1) / 5
    how = "LBAudioDetectiveProcessAudioURLs (one batch call per round, the C call timed)"
    n_sub = sum(f.number_of_subfingerprints for f in fps)
    out = {
        "workload": f"configs[0]: {os.path.basename(a)} vs {os.path.basename(b)} through LBAudioDetectiveCompareAudioURLs "
                    f"(reference defaults 5512 Hz / 2048 / 64, upstream's file loop); then the {len(paths)} bundled fixtures "
                    f"({seconds:.0f} s of audio) x {rounds} through {how}",
        "compare_audio_urls_ms": round(pair_ms, 4), "match": m,
        "files_per_s": round(len(paths) / dt60, 1), "ms_per_call_of_60": round(dt60 * 1e3, 3),
        "audio_seconds_per_s": round(seconds / dt60, 1), "subfingerprints_per_round": n_sub,
        "through_the_python_mirror": {"ms_per_call_of_60": round(dt60_py * 1e3, 3), "files_per_s": round(len(paths) / dt60_py, 1)},
    }
    # the same fixtures a hundred times over in ONE call: what a catalogue build looks like (two-slot pipeline: read + unpack
    # of one run overlaps the device work of the run before), and the same call with the pipeline switched off
    many = paths * 100
    big = det.process_audio_urls(many)
    same = bool(all(big[i].to_bools().tobytes() == fps[i % len(paths)].to_bools().tobytes() for i in range(0, len(many), 7)))
    del big
    dtb = c_call(many, 3, 1)
    det.set_file_pipeline(False)
    dtb_off = c_call(many, 3, 1)
    det.set_file_pipeline(True)
    out["one_call_of_6000_files"] = {
        "files": len(many), "ms": round(dtb * 1e3, 2), "files_per_s": round(len(many) / dtb, 1),
        "us_per_file": round(dtb * 1e6 / len(many), 2), "same_as_the_60_file_call": same,
        "without_the_pipeline_ms": round(dtb_off * 1e3, 2), "pipeline_gain": round(dtb_off / dtb, 3),
    }
    if not args.no_cpu_baseline:
        # the independent oracle end to end (own container reader, IMA4 / LPCM decoder, converter, upstream's file loop), ONE
        # FILE PER THREAD on every usable core (round-4 review: a baseline that runs its files one after the other is a
        # straw man), and one file on one thread
        cfg = O.Config()
        threads = usable_cores()
        sample = (paths * (1 + (2 * threads) // len(paths)))[: max(len(paths), 2 * threads)]       # at least two files per thread
        O.fingerprint_files(paths[:2], cfg, 1, O.TAIL_NOTHING, 0, 2)
        t1 = time.perf_counter()
        got = O.fingerprint_files(sample, cfg, 1, O.TAIL_NOTHING, 0, threads)
        dt_all = time.perf_counter() - t1
        t1 = time.perf_counter()
        O.fingerprint_files(paths[:3], cfg, 1, O.TAIL_NOTHING, 0, 1)
        dt1 = (time.perf_counter() - t1) / 3
        same = all(np.array_equal(got[i], fps[i % len(paths)].to_bools()) for i in range(len(sample)))
        out["cpu_baseline"] = {
            "value": round(len(sample) / dt_all, 2), "unit": "files/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
            "per_core": round(len(sample) / dt_all / threads, 3),
            "single_thread": {"value": round(1.0 / dt1, 3), "unit": "files/s", "cores": 1, "sample": "the first three fixtures, one after the other"},
            "sample": f"the independent oracle end to end (oracle/lbad_file_oracle.c:lbo_fingerprint_files: own container reader, IMA4 / "
                      f"LPCM decoder, converter, then upstream's file loop), {len(sample)} files, one file per OpenMP thread on {threads} "
                      f"threads, {dt_all:.1f} s.  What a file costs a CPU is the reference's own loop: a 2048-point FFT every 8 samples "
                      f"(6 200 of them for nine seconds of audio, 0.6 GFLOP) in front of 385 converter taps per sample",
        }
        out["parity"] = {"files_checked": len(sample), "bit_exact": bool(same)}
    return out

# ---- one rank -----------------------------------------------------------------------------------------
def run_rank(args) -> int:
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with matching values\n")
        return 2
    dry = args.backend == "gloo"
    dist_on = world > 1 or args.force_dist          # every branch below that talks to torch.distributed
    # a free port only where ONE process makes the whole group (--force-dist at one rank): ranks started by hand without
    # MASTER_PORT must agree on a port, and each picking its own free one never meets the others (round-5 advice) -- they
    # keep the fixed default below
    if dist_on and world == 1 and "MASTER_PORT" not in os.environ:
        os.environ["MASTER_PORT"] = str(free_port())
    if dry and args.clips != 0:
        sys.stderr.write("bench.py: --backend gloo is the CPU dry run and needs --clips 0\n")
        return 2
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if dry:
        dev = torch.device("cpu")
        if dist_on:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        if torch.cuda.device_count() < world:
            sys.stderr.write(f"bench.py: {world} ranks need {world} GPUs, {torch.cuda.device_count()} visible\n")
            return 3
        torch.cuda.set_device(local_rank if world > 1 else 0)
        dev = torch.device("cuda", torch.cuda.current_device())
        if dist_on:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    n_gpus = world
    rccl_ranks = dist.get_world_size() if (dist_on and dist.is_initialized()) else 1
    me = {"rank": rank, "local_rank": local_rank, "device": (str(dev)),
          "name": (torch.cuda.get_device_name(dev) if not dry else "cpu"), "pid": os.getpid()}
    if not dry:
        # what a first multi-GPU run needs to see per rank (round 6): which device the rank really sits on, and how the chip
        # is partitioned -- 256 CUs = the eight XCDs as one device (SPX), 32 = one XCD per device (CPX): the kernels' XCD-aware
        # claim order (blockIdx & 7) and one-workgroup-per-CU grids assume the first and only lose speed on the second
        prop = torch.cuda.get_device_properties(dev)
        cus = int(prop.multi_processor_count)
        me.update({"device_index": int(torch.cuda.current_device()), "compute_units": cus, "xcds": max(1, cus // 32),
                   "partition": "SPX (one device = the whole chip)" if cus >= 256 else f"{cus} CUs: a partition of the chip",
                   "arch": getattr(prop, "gcnArchName", "?"), "hbm_GB": round(prop.total_memory / 1e9, 1)})
    devices = [me]

    def regather_devices():
        """the ranks' descriptions again (after the library's communicator was made: comm_count / comm_rank)"""
        nonlocal devices
        if dist_on:
            devices = [None] * world
            dist.all_gather_object(devices, me)
        else:
            devices = [me]
        return devices

    regather_devices()

    from lbaudiodetective_amd import sharded
    sharded.FORCE_COLLECTIVES = bool(args.force_dist)

    def barrier():
        if not dry:
            torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    result = {
        "metric": "audio_seconds_fingerprinted_per_sec", "value": None, "unit": "audio-s/s", "n_gpus": n_gpus,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "rccl_ranks": rccl_ranks, "backend": args.backend, "devices": devices,
    }
    if args.force_dist:
        result["force_dist"] = ("torch.distributed initialised at this world size and every collective branch taken: "
                                "init_process_group, all_gather_object, MAX all-reduces (passes, elapsed, stats), "
                                "broadcast_object_list of the communicator id, all_reduce(MIN) of the communicator check, "
                                "barrier, destroy_process_group")
    if dry:
        result["dry_run"] = "CPU dry run of launcher, rendezvous, shard arithmetic and key reduction; no kernel ran"

    # =========================== fingerprint leg (the headline) ===========================================
    if args.clips > 0:
        import lbaudiodetective_amd as lb
        det = lb.Detective().configure(sample_rate=RATE, window=WINDOW, stride=STRIDE)
        det.set_kernel_variant(args.variant)
        per = det.subfingerprint_count(SAMPLES)
        n_clips = args.clips
        clips = torch.empty((n_clips, SAMPLES), dtype=torch.float32, device=dev)
        lb.synth_clips_device(SEED, rank * n_clips, n_clips, RATE, SAMPLES, out=clips)   # untimed, resident
        packed = torch.empty((n_clips, per, lb.PACKED_BYTES), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()

        def one_pass():
            det.fingerprint_clips_device(clips, out=packed)

        # size a step: two calibration passes (untimed), then as many passes per step as --min-seconds needs
        one_pass()
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        one_pass()
        one_pass()
        torch.cuda.synchronize()
        pass_s = (time.perf_counter() - c0) / 2
        passes = args.passes_per_step or max(1, int(-(-args.min_seconds // (args.steps * pass_s))))
        if dist_on:                                         # every rank must run the same number of passes
            t = torch.tensor([passes], dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            passes = int(t.item())

        def step():
            for _ in range(passes):
                one_pass()

        for _ in range(args.warmup):
            step()
        barrier()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        tele = Telemetry(lb, torch) if rank == 0 else None
        if tele:
            tele.start()
        det.set_stage_timing(True)      # HIP events around each kernel, on the launch stream, inside the timed region
        t0 = time.perf_counter()
        for s in range(args.steps):
            ev[s][0].record()
            step()
            ev[s][1].record()
        barrier()
        elapsed = time.perf_counter() - t0
        if tele:
            tele.stop()
        if dist_on:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        ms_per_step = elapsed * 1e3 / args.steps
        value = n_gpus * n_clips * passes * args.steps / elapsed       # audio-seconds per second, whole job
        if dist_on:
            # the OPTIONAL gather of sharded fingerprinting (160 B per clip; the path itself needs no collective), untimed:
            # every rank's first clips in rank order, checked against what this rank holds
            head = min(16, n_clips)
            allp = sharded.gather_packed(packed[:head])
            assert allp.shape[0] == head * world and torch.equal(allp[rank * head:(rank + 1) * head], packed[:head])
            result["gather_packed"] = {"clips_per_rank": head, "gathered": int(allp.shape[0])}
        step_ms = [a.elapsed_time(b) for a, b in ev]
        kern_avg_ms = sum(step_ms) / len(step_ms) / passes              # one pass, HIP events on the launch stream
        stage1_ms, stage2_ms, launches = det.stage_times()      # summed over the timed passes
        det.set_stage_timing(False)
        launches_per_pass = launches // (args.steps * passes)
        rows_ms = stage1_ms / launches                          # dominant kernel: average launch duration
        clips_per_launch = n_clips / launches_per_pass
        qn = max(1, len(step_ms) // 4)
        sustained = {
            "timed_region_s": round(elapsed, 3), "passes_per_step": passes, "passes_timed": passes * args.steps,
            "ms_per_pass": round(ms_per_step / passes, 4),
            "step_ms_first_quartile": round(sum(step_ms[:qn]) / qn, 4),
            "step_ms_last_quartile": round(sum(step_ms[-qn:]) / qn, 4),
            "step_ms_min": round(min(step_ms), 4), "step_ms_max": round(max(step_ms), 4),
        }
        if tele:
            sustained.update(tele.summary())

        if rank == 0:
            alg_bytes = algorithmic_bytes_per_clip(SAMPLES, WINDOW, STRIDE)
            canon_flops = per * 128 * 2.5 * WINDOW * 10            # 2.5 W log2 W per window
            traffic, traffic_src = profile_traffic(args.variant, clips_per_launch)
            achieved = alg_bytes * clips_per_launch / (rows_ms * 1e-3) / 1e9
            result.update({
                "value": round(value, 1), "ms_per_step": round(ms_per_step, 4),
                "ms_per_pass": round(ms_per_step / passes, 4), "passes_per_step": passes, "sustained": sustained,
                "config": {
                    "workload": "configs[1]: 100k synthetic 1 s @44.1 kHz mono clips, 1024-pt FFT, stride 64, "
                                f"fingerprint-only, input resident in HBM; one step = {passes} passes over the batch",
                    "clips_per_gpu": n_clips, "samples_per_clip": SAMPLES, "window": WINDOW, "stride": STRIDE,
                    "bands": 32, "subfingerprints_per_clip": per, "kernel_variant": args.variant,
                    "parallelism": f"clips sharded x{n_gpus}, no collective",
                },
                "per_gpu_value": round(value / n_gpus, 1),
                "roofline": {
                    # 93 canonical flop per algorithmic byte against a ridge of ~20 -- the kernel is
                    # bound by FP32 VALU issue and LDS latency, not by HBM.  achieved / peak / frac are the HBM-side
                    # figures the contract asks for (algorithmic bytes / launch duration); the VALU figures follow.
                    "bound": "valu",
                    "kernel": "stage 1, windows -> frame rows (frame_rows_pruned_kernel when the pruned FFT applies, "
                              "else fft_bands_kernel): dominant kernel of the pass",
                    "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5),
                    "traffic": traffic, "traffic_source": traffic_src,
                    "algorithmic_bytes_per_clip": alg_bytes, "clips_per_launch": clips_per_launch,
                    "launches_per_pass": launches_per_pass, "kernel_ms_avg": round(rows_ms, 4),
                    "stage2_kernel_ms_avg": round(stage2_ms / launches, 4), "pass_ms_avg": round(kern_avg_ms, 4),
                    "fp32_canonical_tflops": round(canon_flops * clips_per_launch / (rows_ms * 1e-3) / 1e12, 3),
                    "fp32_peak_tflops": FP32_PEAK_TFLOPS,
                    "fp32_frac_canonical": round(canon_flops * clips_per_launch / (rows_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4),
                    **executed_fields(RATE, WINDOW, canon_flops * clips_per_launch / (rows_ms * 1e-3) / 1e12, args.variant),
                    "whole_pass_achieved_GBps": round(alg_bytes * n_clips / (kern_avg_ms * 1e-3) / 1e9, 2),
                },
            })
            # parity of the bench's own data against the oracle (untimed, the oracle is the checker)
            from oracle import oracle as O
            cfg = O.Config(RATE, WINDOW, STRIDE)
            n_par = min(64, n_clips)
            host = clips[:n_par].cpu().numpy()
            want = O.fingerprint_batch(host, cfg, nthreads=min(8, usable_cores()))
            got = lb.unpack_packed(packed[:n_par].cpu().numpy(), 200).reshape(n_par, per, 200)
            result["parity"] = {"clips_checked": n_par, "bit_exact": bool(np.array_equal(got, want))}

            if not args.no_cpu_baseline and world == 1:
                threads = usable_cores()
                n_cpu = args.cpu_sample or min(n_clips, 3000 * threads)     # ~9 s at ~2.8 ms per clip per thread
                sample = clips[:n_cpu].cpu().numpy()
                O.fingerprint_batch(sample[: 2 * threads], cfg, nthreads=threads)    # warm the caches/threads
                t1 = time.perf_counter()
                O.fingerprint_batch(sample, cfg, nthreads=threads)
                dt = time.perf_counter() - t1
                n_one = min(n_cpu, 1200)                                    # ~3.5 s on one core
                t1 = time.perf_counter()
                O.fingerprint_batch(sample[:n_one], cfg, nthreads=1)
                dt1 = time.perf_counter() - t1
                result["cpu_baseline"] = {
                    "value": round(n_cpu / dt, 2), "unit": "audio-s/s", "cores": threads, "kind": "port",
                    "per_core": round(n_cpu / dt / threads, 2),
                    "single_thread": {"value": round(n_one / dt1, 2), "unit": "audio-s/s", "cores": 1,
                                      "sample": f"first {n_one} clips, one thread, {dt1:.1f} s"},
                    "cpu_model": cpu_model(), "host_logical_cpus": os.cpu_count(),
                    "sample": f"first {n_cpu} clips of the same batch through oracle/lbad_oracle.c "
                              f"(scalar radix-2 restatement of the reference, not vDSP: Accelerate does not exist on "
                              f"Linux), {threads} OpenMP threads, {dt:.1f} s",
                }
        del clips, packed
        # ---- the other processing configurations of BASELINE.json (side measurements, one GPU) ---------------
        if world == 1 and not args.no_other_configs:
            from oracle import oracle as O
            others = {}
            for key, rate, window, seconds, n_o, stereo in (("configs0_settings", 5512, 2048, 9, 20_000, False),
                                                            ("configs4", 48000, 4096, 1, 10_000, True)):
                d2 = lb.Detective().configure(sample_rate=rate, window=window)
                samples = rate * seconds
                c2 = lb.synth_clips_device(SEED, 0, n_o, rate, samples, stereo)
                p2 = d2.fingerprint_clips_device(c2)
                torch.cuda.synchronize()
                d2.set_stage_timing(True)
                for _ in range(3):                              # calls_o below
                    d2.fingerprint_clips_device(c2, out=p2)
                s1, s2, ln = d2.stage_times()
                d2.set_stage_timing(False)
                per2 = int(p2.shape[1])
                ab = algorithmic_bytes_per_clip(samples, window, STRIDE)
                calls_o = 3
                (ms1, ms2, lpc), (k1, k2) = per_call_stage_times(s1, s2, ln, calls_o)   # one CALL over the n_o clips
                clips_per_launch_o = n_o / lpc
                canon = per2 * 128 * 2.5 * window * (window.bit_length() - 1)          # per clip, SURVEY 8d
                ach = ab * clips_per_launch_o / (k1 * 1e-3) / 1e9
                want = O.fingerprint_batch(c2[:4].cpu().numpy(), O.Config(rate, window), nthreads=4)
                got = lb.unpack_packed(p2[:4].cpu().numpy(), 200).reshape(4, per2, 200)
                traffic_o, traffic_src_o = profile_traffic(0, clips_per_launch_o, "stage1_stream_2048" if window == 2048 else "stage1_stream_4096")
                others[key] = {
                    "workload": f"{n_o} clips x {seconds} s @ {rate} Hz{' stereo-summed' if stereo else ''}, {window}-pt FFT, stride 64",
                    "calls_timed": calls_o, "launches_per_call": lpc,
                    "stage1_ms": round(ms1, 3), "stage2_ms": round(ms2, 3),
                    "audio_seconds_per_s": round(n_o * seconds / ((ms1 + ms2) * 1e-3), 1),
                    "clips_per_s": round(n_o / ((ms1 + ms2) * 1e-3), 1),
                    "roofline": {
                        "bound": "valu", "kernel": "stage 1 (rows_stream2_kernel at 2048-sample windows, rows_stream_kernel at 4096)",
                        "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                        "traffic": traffic_o, "traffic_source": traffic_src_o,
                        "algorithmic_bytes_per_clip": ab, "clips_per_launch": clips_per_launch_o,
                        "kernel_ms_avg": round(k1, 4), "stage2_kernel_ms_avg": round(k2, 4),
                        "fp32_canonical_tflops": round(canon * clips_per_launch_o / (k1 * 1e-3) / 1e12, 2),
                        "fp32_peak_tflops": FP32_PEAK_TFLOPS,
                        "fp32_frac_canonical": round(canon * clips_per_launch_o / (k1 * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4),
                        **executed_fields(rate, window, canon * clips_per_launch_o / (k1 * 1e-3) / 1e12, 0),
                    },
                    "parity": {"clips_checked": 4, "bit_exact": bool(np.array_equal(got, want))},
                }
                del c2, p2
            if rank == 0:
                result["other_configs"] = others
    else:
        result["config"] = {"workload": "fingerprint leg skipped (--clips 0)", "parallelism": f"x{n_gpus}"}

    # =========================== compare leg (side measurement) ===========================================
    total = args.corpus if args.corpus >= 0 else (1_000_000 if world == 1 else 10_000_000)
    if total > 0:
        per = 5
        begin, end = sharded.shard_range(total, rank, world)
        planted = (PLANTED_1GPU if world == 1 else PLANTED_SHARDED) % total
        planted_rank = next(r for r in range(world) if sharded.shard_range(total, r, world)[0] <= planted < sharded.shard_range(total, r, world)[1])
        reps = 20
        if dry:
            # no scan: every rank contributes the key its shard would produce if its best entry were its first one
            # at a chance-level score, the planted rank the planted entry at 0.93; exercises make_key / MAX / decode
            def fbits(x):
                return struct.unpack("<I", struct.pack("<f", x))[0]
            local = sharded.make_key(fbits(0.5 + 0.001 * rank), begin) if end > begin else 0
            if begin <= planted < end:
                local = sharded.make_key(fbits(0.93), planted)
            key = torch.tensor([local], dtype=torch.int64)
            t1 = time.perf_counter()
            for _ in range(reps):
                k2 = key.clone()
                sharded.allreduce_best(k2)
            ar_ms = (time.perf_counter() - t1) * 1e3 / reps
            best = sharded.decode_key(int(k2.item()))
            if rank == 0:
                result["compare"] = {
                    "workload": f"DRY RUN: 1 query vs {total} fingerprints, sharded x{world} (no scan)",
                    "best_index": best[0], "best_score": best[1], "planted_index": planted, "planted_rank": planted_rank,
                    "entries_per_rank": [sharded.shard_range(total, r, world)[1] - sharded.shard_range(total, r, world)[0] for r in range(world)],
                    "allreduce_ms": round(ar_ms, 4), "collective": f"{args.backend} all_reduce(MAX) of one int64",
                }
        else:
            import lbaudiodetective_amd as lb
            # the exchange step runs INSIDE the library: its own RCCL communicator (ncclCommInitRank through
            # LBAudioDetectiveCommInitRank; torch.distributed only carries the 128-byte id) and
            # LBAudioDetectiveCorpusQuerySharded = scan + ncclAllReduce(ncclUint64, ncclMax) + 8-byte read-back
            comm, comm_note = None, None
            try:
                comm = quiet_stdout(lambda: sharded.make_comm(rank, world))   # RCCL prints a version banner on stdout
            except Exception as e:                          # noqa: BLE001 -- keep the scaling run alive, say what happened
                comm_note = f"LBAudioDetectiveCommInitRank failed ({e}); keys reduced through torch.distributed instead"
                sys.stderr.write("bench.py: " + comm_note + "\n")
            # the library's communicator as RCCL itself describes it: ncclCommCount must be the world size on EVERY rank
            me["comm_count"], me["comm_rank"] = (comm.info() if comm is not None else (0, -1))
            result["devices"] = regather_devices()
            if dist_on:                                     # all ranks on the same path, or none
                ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=dev)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) == 0 and comm is not None:
                    comm.dispose()
                    comm, comm_note = None, "another rank could not create the library's communicator; torch.distributed instead"
            sc = lb.ShardedCorpus(200, per, total, rank=rank, world_size=world, comm=comm)
            chunk = 1 << 20
            for b in range(sc.begin, sc.end, chunk):
                sc.append_packed_device(lb.synth_corpus_device(CSEED, b, min(chunk, sc.end - b), per, 200))
            qsrc = lb.unpack_packed(lb.synth_corpus_device(CSEED, planted, 1, per, 200).cpu().numpy(), 200)
            rng = np.random.default_rng(7)                      # same query on every rank
            flip = rng.random((per, 100)) < 0.07
            q = qsrc.copy()
            q[:, 0::2] = np.where(flip, qsrc[:, 1::2], qsrc[:, 0::2])
            q[:, 1::2] = np.where(flip, qsrc[:, 0::2], qsrc[:, 1::2])
            fq = lb.Fingerprint.from_bools(q)
            if dist_on:                                         # the 160-byte query broadcast of the sharded compare
                fq = sharded.broadcast_fingerprint(fq if rank == 0 else None, src=0)
                assert np.array_equal(fq.to_bools(), q)
            key = torch.zeros(1, dtype=torch.int64, device=dev)
            for _ in range(3):
                best = sc.query(fq)
            barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                sc.local.query_key_device(fq, key, 0, index_base=sc.begin)       # the local scan alone
            e1.record()
            torch.cuda.synchronize()
            scan_ms = e0.elapsed_time(e1) / reps
            local_key = key.clone()
            ar_ms = None
            if dist_on:                                         # for reference: 8 bytes through torch.distributed's RCCL
                barrier()
                t1 = time.perf_counter()
                for _ in range(reps):
                    k2 = local_key.clone()
                    sharded.allreduce_best(k2)
                    torch.cuda.synchronize()
                ar_ms = (time.perf_counter() - t1) * 1e3 / reps
            barrier()
            t1 = time.perf_counter()
            for _ in range(reps):
                best = sc.query(fq)                             # scan + ncclAllReduce + 8-byte read-back, in the library
            lat_ms = (time.perf_counter() - t1) * 1e3 / reps
            api_lat_ms = None
            if world == 1:
                # the host-synchronous drop-in call (LBAudioDetectiveCorpusQuery): one launch, result polled in pinned memory
                for _ in range(3):
                    best_api = sc.local.query(fq)
                t1 = time.perf_counter()
                for _ in range(reps * 5):
                    best_api = sc.local.query(fq)
                api_lat_ms = (time.perf_counter() - t1) * 1e3 / (reps * 5)
                assert best_api == best, (best_api, best)
            stats = torch.tensor([scan_ms, ar_ms or 0.0, lat_ms], dtype=torch.float64, device=dev)
            if dist_on:
                dist.all_reduce(stats, op=dist.ReduceOp.MAX)
            scan_ms, ar_max, lat_ms = (float(v) for v in stats.tolist())
            n_local_max = -(-total // world)
            if rank == 0:
                result["compare"] = {
                    "workload": (f"configs[2]: 1 query vs {total} fingerprints ({per} x 200 Booleans), one GPU" if world == 1 else
                                 f"configs[3]: 1 query vs {total} fingerprints ({per} x 200 Booleans) in contiguous shards of "
                                 f"{total // world} per rank x{world}, RCCL all-reduce(MAX) of the (score, ~index) key"),
                    "best_index": best[0], "best_score": best[1], "planted_index": planted, "planted_rank": planted_rank,
                    "found_planted": bool(best[0] == planted),
                    "scan_ms": round(scan_ms, 4), "allreduce_ms": (round(ar_max, 4) if dist_on else None),
                    "query_latency_ms": round(api_lat_ms if api_lat_ms is not None else lat_ms, 4),
                    "query_latency_sharded_path_ms": round(lat_ms, 4),
                    "collective_fallback": comm is None,
                    "collective": (comm_note if comm is None else
                                   "ncclAllReduce(count 1, ncclUint64, ncclMax) inside LBAudioDetectiveCorpusQuerySharded, "
                                   f"communicator of {world} rank(s) from ncclCommInitRank; allreduce_ms is the same 8 bytes "
                                   "through torch.distributed, for reference"),
                    "entries_per_s": round(total / ((api_lat_ms if api_lat_ms is not None else lat_ms) * 1e-3), 1),
                    "scan_entries_per_s": round(total / (scan_ms * 1e-3), 1),
                    "achieved_GBps_algorithmic_per_gpu": round(25 * per * n_local_max / (scan_ms * 1e-3) / 1e9, 2),
                    "scan_frac_of_hbm_peak_per_gpu": round(25 * per * n_local_max / (scan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    "scan_reads_from": (f"the Infinity Cache, not HBM: a shard's planes ({sc.local.entry_stride_bytes * n_local_max / 1e6:.0f} MB) fit its "
                                        "256 MB -- the HBM-resident figure is compare_hbm.roofline"
                                        if sc.local.entry_stride_bytes * n_local_max < 256e6 else "HBM"),
                }
                if world == 1 and not args.no_cpu_baseline:
                    # the same query over the same corpus through the oracle's Boolean-per-byte loop
                    # (the reference's layout: 1000 B per entry), all usable cores; also a full-size parity check
                    from oracle import oracle as O
                    threads = usable_cores()
                    host = np.empty((total, per, 200), np.uint8)
                    for b in range(0, total, 1 << 18):
                        m = min(1 << 18, total - b)
                        host[b:b + m] = lb.unpack_packed(
                            lb.synth_corpus_device(CSEED, b, m, per, 200).cpu().numpy(), 200).reshape(m, per, 200)
                    O.corpus_best(q, host[: 1 << 14], 200, nthreads=threads)
                    t1 = time.perf_counter()
                    ci, cs = O.corpus_best(q, host, 200, nthreads=threads)
                    dt = time.perf_counter() - t1
                    n_1 = min(total, 250_000)
                    t1 = time.perf_counter()
                    O.corpus_best(q, host[:n_1], 200, nthreads=1)
                    dt1 = time.perf_counter() - t1
                    # SURVEY 8(d)'s second baseline: the packed popcount scan (4 x 64-bit words per sub-fingerprint,
                    # 160 B per entry), all cores and one; packing is untimed, like the GPU corpus build
                    qw, cw = O.pack_bools(q), O.pack_bools(host)
                    del host
                    O.corpus_best_packed(qw, cw[: 1 << 14], 200, 200, nthreads=threads)
                    t1 = time.perf_counter()
                    pi, ps = O.corpus_best_packed(qw, cw, 200, 200, nthreads=threads)
                    dtp = time.perf_counter() - t1
                    t1 = time.perf_counter()
                    O.corpus_best_packed(qw, cw, 200, 200, nthreads=1)
                    dtp1 = time.perf_counter() - t1
                    del cw
                    result["compare"]["cpu_baseline"] = {
                        "value": round(total / dt, 1), "unit": "entries/s", "cores": threads, "kind": "port",
                        "single_thread": {"value": round(n_1 / dt1, 1), "unit": "entries/s", "cores": 1,
                                          "sample": f"first {n_1} entries, {dt1 * 1e3:.0f} ms"},
                        "sample": f"all {total} entries as {per} x 200 Booleans (the reference's layout, 1000 B per entry) through "
                                  f"oracle/lbad_oracle.c:lbo_corpus_best, {threads} OpenMP threads, {dt * 1e3:.0f} ms",
                        "cpu_model": cpu_model(),
                        "packed_popcount": {
                            "value": round(total / dtp, 1), "unit": "entries/s", "cores": threads, "kind": "port",
                            "GBps": round(32 * per * total / dtp / 1e9, 2),
                            "single_thread": {"value": round(total / dtp1, 1), "unit": "entries/s", "cores": 1,
                                              "GBps": round(32 * per * total / dtp1 / 1e9, 2)},
                            "agrees_with_boolean_loop": bool(pi == ci and np.float32(ps).view(np.uint32) == np.float32(cs).view(np.uint32)),
                            "sample": f"all {total} entries as {per} x 4 64-bit words (160 B per entry) through "
                                      f"oracle/lbad_oracle.c:lbo_corpus_best_packed (two popcounts per word), {threads} OpenMP "
                                      f"threads {dtp * 1e3:.1f} ms, one thread {dtp1 * 1e3:.1f} ms; packing untimed",
                        },
                    }
                    result["compare"]["parity"] = {
                        "entries_checked": total,
                        "bit_exact": bool(ci == best[0] and np.float32(cs).view(np.uint32) == np.float32(best[1]).view(np.uint32)),
                    }

            # HBM-resident scan on one GPU: the 1 M corpus (128 MB) fits the 256 MiB Infinity Cache, this one does not
            if args.corpus_hbm > 0 and world == 1:
                n_big = args.corpus_hbm
                big = lb.Corpus(200, per, n_big)
                for b in range(0, n_big, 1 << 20):
                    big.append_packed_device(lb.synth_corpus_device(CSEED, b, min(1 << 20, n_big - b), per, 200))
                for _ in range(3):
                    big.query_key_device(fq, key)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(10):
                    big.query_key_device(fq, key)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 10
                idx, score = lb.Corpus.decode_key(int(key.item()))
                result["compare_hbm"] = {
                    "workload": f"1 query vs {n_big} fingerprints on one GPU ({big.entry_stride_bytes * n_big / 1e9:.2f} GB, HBM-resident)",
                    "scan_ms": round(ms, 4), "best_index": idx, "best_score": score,
                    "roofline": {"bound": "hbm", "achieved": round(25 * per * n_big / (ms * 1e-3) / 1e9, 1),
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(25 * per * n_big / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                 "layout_GBps": round(big.entry_stride_bytes * n_big / (ms * 1e-3) / 1e9, 1)},
                }
                # Q x N on the uniform corpus (LBAudioDetectiveTests.m:57-91): eight queries share ONE pass over the planes;
                # every key must equal the single query's
                fps8 = []
                for k8 in range(8):
                    src8 = lb.unpack_packed(lb.synth_corpus_device(CSEED, (1_234_567 * (k8 + 1)) % n_big, 1, per, 200).cpu().numpy(), 200)
                    fps8.append(lb.Fingerprint.from_bools(src8.reshape(per, 200)))
                keys8 = torch.zeros(8, dtype=torch.int64, device=dev)
                single8 = torch.zeros(8, dtype=torch.int64, device=dev)
                for i8, f8 in enumerate(fps8):
                    big.query_key_device(f8, single8[i8:i8 + 1])
                for _ in range(3):
                    big.query_batch_keys_device(fps8, keys8)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(10):
                    big.query_batch_keys_device(fps8, keys8)
                e1.record()
                torch.cuda.synchronize()
                ms8 = e0.elapsed_time(e1) / 10
                result["compare_hbm"]["eight_queries_one_call"] = {
                    "one_call_ms": round(ms8, 4), "times_one_query": round(ms8 / ms, 2), "same_keys": bool(torch.equal(keys8, single8)),
                    "kernel": "compare_planes_batch_kernel<5> (round 6: two three-input operations and a count per word, the quotient by "
                              "fused multiply-adds; vector-ALU-bound, DESIGN 9.8)",
                    "subfingerprint_compares_per_s": round(8 * per * n_big / (ms8 * 1e-3), 1)}
                big.dispose()

    # =========================== sliding compare on a ragged corpus (side measurement) ====================
    if world == 1 and not dry and not args.no_sliding and args.corpus != 0:
        result["compare_sliding"] = sliding_leg(args, torch, np)

    # =========================== the reference's own workload: files (BASELINE configs[0]) ================
    if world == 1 and not dry and not args.no_files:
        files = files_leg(args, torch, np)
        if files:
            result["configs0_files"] = files

    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    rc = 0
    if rank == 0:
        bad = self_check(result)
        result["self_check"] = {"ok": not bad, "violations": bad}
        if bad:
            sys.stderr.write("bench.py: SELF-CHECK FAILED: " + "; ".join(bad) + "\n")
            rc = 4
        wrong = [d for d in result.get("devices", []) if "comm_count" in d and (d["comm_count"] != world or d["comm_rank"] != d["rank"])]
        if wrong:
            sys.stderr.write("bench.py: the library's RCCL communicator does not span the run: " + "; ".join(
                f"rank {d['rank']} (device {d.get('device_index')}, pid {d['pid']}) reports {d['comm_count']} rank(s), "
                f"itself as rank {d['comm_rank']}" for d in wrong) + f" -- expected {world}\n")
            rc = rc or 6
        result["library_comm_count"] = (sorted({d["comm_count"] for d in result.get("devices", []) if "comm_count" in d}) or [None])[0] \
            if len({d.get("comm_count") for d in result.get("devices", []) if "comm_count" in d}) <= 1 else "differs between ranks"
        if dist_on and result.get("compare", {}).get("collective_fallback"):
            sys.stderr.write("bench.py: the library's RCCL communicator could not be created; the compare leg fell back to "
                             "torch.distributed -- a scaling run must not pass like this\n")
            rc = rc or 5
        print(json.dumps(result), flush=True)
    return rc


def main() -> int:
    args = parse_args()
    if args.gpus < 1:
        sys.stderr.write("bench.py: --gpus must be >= 1\n")
        return 2
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
