#!/usr/bin/env python3
"""Static instruction mix of one kernel of the device library (cross-compiles for gfx950: no GPU needed).

    python tools/isa_summary.py <demangled-name-substring> [--dump out.s] [--asm existing.s] [extra hipcc flags ...]

e.g.  python tools/isa_summary.py 'k_wf_shade<double, false, true, false, false>'
Prints the number of instructions by class (f64 arithmetic, f64 division / sqrt helper instructions, f32, integer,
memory, scalar, branches) of the FIRST kernel whose demangled name contains the substring, plus its labelled basic
blocks with more than --min instructions.  Static counts: loops and divergence are not weighted.
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "rust_raytracer_amd", "csrc", "rt_kernels.hip")


def classify(op: str) -> str:
    if op.startswith(("v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")):
        return "f64 div/sqrt helper"
    if op.startswith("v_") and "_f64" in op:
        return "f64 arith"
    if op.startswith(("v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_mad_i64_i32", "v_mul_hi_i32", "v_mul_lo_i32")):
        return "int mul (quarter rate)"
    if op.startswith("v_") and "_f32" in op:
        return "f32"
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "vmem load"
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store")):
        return "vmem store"
    if op.startswith(("global_atomic", "flat_atomic", "buffer_atomic")):
        return "vmem atomic"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("v_cndmask", "v_mov", "v_accvgpr", "v_readlane", "v_readfirstlane", "v_writelane", "v_perm", "v_bfe", "v_swap")):
        return "valu move/select"
    if op.startswith("v_cmp"):
        return "valu compare"
    if op.startswith("v_"):
        return "valu int/other"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_call")):
        return "branch"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("--asm", default="")
    ap.add_argument("--dump", default="")
    ap.add_argument("--min", type=int, default=0, help="also list basic blocks with at least this many instructions")
    a, extra = ap.parse_known_args()
    if a.asm:
        text = open(a.asm).read()
    else:
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, "k.s")
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-function",
                   "--cuda-device-only", "-S", SRC, "-o", out] + extra
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                sys.stderr.write(r.stderr)
                raise SystemExit(r.returncode)
            text = open(out).read()
    syms = re.findall(r"^(_Z[\w$.]+):\s*(?:;.*)?$", text, flags=re.M)
    dem = subprocess.run(["c++filt"] + syms, capture_output=True, text=True).stdout.splitlines()
    pick = None
    for s, d in zip(syms, dem):
        if a.name in d:
            pick = (s, d)
            break
    if not pick:
        raise SystemExit(f"no kernel matching {a.name!r}; have e.g. {dem[:5]}")
    sym, d = pick
    start = text.index(f"\n{sym}:") + 1
    end = text.index(".Lfunc_end", start)
    end = text.index("\n", end)
    body = text[start:end]
    if a.dump:
        open(a.dump, "w").write(body)
    counts = collections.Counter()
    blocks = []
    cur, n = sym, 0
    for line in body.splitlines():
        s = line.strip()
        if not s or s.startswith((";", "//")):
            continue
        m = re.match(r"^([.\w$]+):", s)
        if m:
            blocks.append((cur, n))
            cur, n = m.group(1), 0
            continue
        if s.startswith("."):
            continue
        op = s.split()[0]
        counts[classify(op)] += 1
        n += 1
    blocks.append((cur, n))
    total = sum(counts.values())
    print(d)
    for k, v in counts.most_common():
        print(f"  {k:26s} {v:6d}  {100.0 * v / total:5.1f} %")
    print(f"  {'total':26s} {total:6d}")
    m = re.search(re.escape(sym) + r"[\s\S]*?\.vgpr_count:\s*(\d+)", text[end:])
    meta = re.search(r"\.name:\s+" + re.escape(sym) + r"\n[\s\S]*?\.vgpr_count:\s*(\d+)", text)
    if meta:
        print("  vgpr_count", meta.group(1))
    if a.min:
        for name, n in blocks:
            if n >= a.min:
                print(f"  block {name:20s} {n}")


if __name__ == "__main__":
    main()
