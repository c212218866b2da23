"""CPU checks of the COMPILED scan kernels (hipcc cross-compiles gfx950 without a GPU): what the round-6 changes rely on is
asserted on the instruction stream itself, not on a comment -- no register of the ragged-corpus scans is spilled, and no
v_bitop3 of the kernels whose query words live in vector registers reads three registers of one bank (register number mod 4:
tools/ubench/operand_rates.hip measured 4 cycles instead of 2 for exactly that case on the MI355X)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def sliding_isa(tmp_path_factory):
    if not os.path.exists(HIPCC) and shutil.which("hipcc") is None:
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa") / "k_sliding.s"
    src = os.path.join(ROOT, "lbaudiodetective_amd", "csrc", "k_sliding.hip")
    # the flags of lbaudiodetective_amd/csrc/Makefile
    cmd = [HIPCC if os.path.exists(HIPCC) else "hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
           "-fhip-fp32-correctly-rounded-divide-sqrt", "-x", "hip", "--cuda-device-only", "-S", "-I" + os.path.join(ROOT, "include"),
           src, "-o", str(out)]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    return open(out).read()


def kernels(isa):
    """name -> (text of the function, metadata block)"""
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", isa):
        meta[m.group(1)] = int(m.group(2))
    bodies = {}
    for m in re.finditer(r"^(_ZN4lbad\S+):.*?\n(.*?)s_endpgm", isa, re.M | re.S):
        bodies[m.group(1)] = m.group(2)
    return meta, bodies


def test_scan_kernels_spill_nothing(sliding_isa):
    meta, _ = kernels(sliding_isa)
    scans = {k: v for k, v in meta.items() if "compare_sliding_kernel" in k or "compare_short_multi_kernel" in k}
    assert len(scans) >= 8 + 21, sorted(scans)
    assert {k: v for k, v in scans.items() if v != 0} == {}


def test_no_three_sources_on_one_register_bank(sliding_isa):
    _, bodies = kernels(sliding_isa)
    checked = 0
    for name, text in bodies.items():
        multi = "compare_short_multi_kernel" in name
        # compare_sliding_kernel<FULL, ALL_FEED, QLDS, QN, THREADS>: the instances that read the query from LDS (QLDS = 1)
        lds_query = re.search(r"compare_sliding_kernelILb[01]ELb[01]ELb1E", name) is not None
        if not (multi or lds_query):
            continue
        ops = bad = 0
        for line in text.splitlines():
            if "v_bitop3_b32" not in line:
                continue
            regs = [int(x) for x in re.findall(r"\bv(\d+)\b", line.split("bitop3:")[0])][1:]
            ops += 1
            if len(regs) == 3 and len(set(regs)) == 3 and len({r % 4 for r in regs}) == 1:
                bad += 1
        assert ops > 40, (name, ops)
        assert bad == 0, (name, ops, bad)
        checked += 1
    assert checked >= 6 + 21


def test_ratio_without_the_table_is_the_ieee_quotient(tmp_path):
    """compare_short_multi_kernel computes hits / possible as fma(hf, rh, RN(hf * rl)) with rh = RN(1 / pf), rl = RN(fma(-pf, rh, 1) * rh)
    instead of reading the table of 5151 quotients: tools/verify_ratio_fma.c runs exactly those correctly rounded operations for
    every 0 <= hits <= possible <= 100 and compares the bits with (float)hits / (float)possible (LBAudioDetectiveFingerprint.m:175)."""
    if shutil.which("gcc") is None:
        pytest.skip("no host compiler")
    exe = tmp_path / "verify_ratio_fma"
    build = subprocess.run(["gcc", "-O2", "-mfma", "-ffp-contract=off", os.path.join(ROOT, "tools", "verify_ratio_fma.c"), "-o", str(exe), "-lm"],
                           capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert run.returncode == 0 and "5151 pairs checked, 0 disagree" in run.stdout, run.stdout[-500:]
