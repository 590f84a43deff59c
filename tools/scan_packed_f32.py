"""List every PACKED fp32 VALU instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 / v_pk_mov_b32) in the gfx950 code of the
product sources, per kernel and operand-selection form.  The shipped build must contain NONE.

Why (evidence: profiles/r05_gn_exec_repro.jsonl, tools/gn_exec_repro.cpp, tools/probe_pk_swap.hip): on the MI355X a
packed fp32 operation whose LOW result reads the HIGH half of src1 / src2 (`op_sel:[0,0,1] op_sel_hi:[1,1,0]`, `op_sel:[0,1]
op_sel_hi:[1,0]`: what hipcc emits for horizontal adds and lane-pair swaps) returns wrong values in lanes 48..63 while another
wave of the same SIMD has MFMAs in flight -- beside an MFMA loop up to 11 % of those lanes' results; alone, or beside VALU / LDS /
packed-fp16 / AccVGPR-move / empty kernels, never.  That is what made the round-3 GroupNorm statistics lose sum-of-squares terms
beside another stream's generic-kernel launches (the round-4 "EXEC update" reading was wrong: loop forms with wait states in front
of the EXEC update still fail, forms without the cross-half instruction do not).  The library is therefore built with the packed
fp32 instructions switched off (`-target-feature -packed-fp32-ops`, csrc/build.py HIP_FLAGS); this scan is what
tests/test_isa_hazards.py asserts on.  It replaces tools/scan_exec_sites.py (round 4), which looked for the wrong pattern.

    python tools/scan_packed_f32.py [extra hipcc flags ...]      # needs hipcc; ~30 s; exit status 1 if any instruction is found
"""
import collections
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from editanything_amd.csrc import build  # noqa: E402

PACKED = re.compile(r"^v_pk_(fma|mul|add)_f32|^v_pk_mov_b32")


def scan_asm(asm):
    """-> {kernel: Counter({"v_pk_fma_f32 op_sel:[0,0,1] op_sel_hi:[1,1,0]": n, ...})}"""
    out = collections.defaultdict(collections.Counter)
    kern = None
    for line in asm.split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kern = m.group(1)
        t = line.strip()
        if PACKED.match(t):
            form = " ".join([t.split()[0]] + [w for w in t.split() if w.startswith(("op_sel", "neg_"))])
            out[kern][form] += 1
    return out


def cross_half(form):
    """True when a LOW result reads a HIGH half (an op_sel bit is set): the form measured to fail beside MFMAs."""
    m = re.search(r"op_sel:\[([01,]*)\]", form)
    return bool(m and "1" in m.group(1))


def scan_source(path, extra=()):
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        subprocess.check_call([build._hipcc()] + build.HIP_FLAGS + list(extra) + ["--cuda-device-only", "-S", path, "-o", tmp.name],
                              stderr=subprocess.DEVNULL)
        return scan_asm(open(tmp.name).read())


def scan_product(extra=()):
    """{source: {kernel: Counter}} over every source of the product library, compiled with the product flags."""
    srcs = [os.path.join(build.HERE, s) for s in build.SOURCES]
    with ThreadPoolExecutor(len(srcs)) as ex:
        return dict(zip(build.SOURCES, ex.map(lambda p: scan_source(p, extra), srcs)))


def main():
    found = scan_product(sys.argv[1:])
    total = bad = 0
    for src, kernels in found.items():
        n = sum(sum(c.values()) for c in kernels.values())
        nb = sum(v for c in kernels.values() for f, v in c.items() if cross_half(f))
        total += n
        bad += nb
        print(f"{src}: {n} packed fp32 instructions in {len(kernels)} kernels, {nb} with a cross-half source selection")
        for k, c in sorted(kernels.items()):
            print(f"    {'*' if any(cross_half(f) for f in c) else ' '} {str(k)[:100]}  {dict(c)}")
    print(f"total: {total} packed fp32 instructions, {bad} cross-half (* = the form that fails beside MFMAs)")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
