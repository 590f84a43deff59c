"""CPU test (needs hipcc, no GPU): the gfx950 code of the product library contains NO packed fp32 VALU instruction.

Round 5: `v_pk_fma_f32` / `v_pk_mul_f32` with a cross-half source selection -- hipcc's code for horizontal adds
-- return wrong data in lanes 48..63 on the MI355X while another wave of the SIMD has MFMAs in flight (tools/probe_pk_swap.hip,
profiles/r05_gn_exec_repro.jsonl); that, not an EXEC-update hazard, is what corrupted the round-3 GroupNorm statistics beside a
second stream.  The build switches the instruction class off (csrc/build.py HIP_FLAGS); this test compiles every product source
with those flags and fails if an instruction of the class shows up again (a flag lost in a refactor, a new inline-asm block).
It also pins the scanner itself on the failing loop form kept under tools/kernels/ for the reproducer."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="hipcc not available")


def test_the_product_build_contains_no_packed_fp32_instruction():
    import scan_packed_f32 as sp
    found = sp.scan_product()
    assert set(found) == set(sp.build.SOURCES)
    offenders = {(src, k): dict(c) for src, kernels in found.items() for k, c in kernels.items()}
    assert not offenders, f"packed fp32 VALU instructions in the product build: {offenders}"


def test_the_scanner_sees_the_round3_loop_form_when_packed_fp32_is_allowed():
    """Control: the same scan on the round-3 GroupNorm statistics loop built the round-3 way (packed fp32 allowed) finds the
    cross-half v_pk_fma_f32 that tools/gn_exec_repro.cpp shows failing."""
    import scan_packed_f32 as sp
    flags = [f for f in sp.build.HIP_FLAGS if f not in ("-Xclang", "-target-feature", "-packed-fp32-ops")]
    saved = sp.build.HIP_FLAGS
    sp.build.HIP_FLAGS = flags + ["-DEA_GN_STATS_LOOP=1"]
    try:
        found = sp.scan_source(os.path.join(sp.build.HERE, "ea_norm.hip"))
    finally:
        sp.build.HIP_FLAGS = saved
    stats = {k: c for k, c in found.items() if "ea_gn_stats_kernel" in str(k)}
    assert stats and any(sp.cross_half(f) for c in stats.values() for f in c), stats
