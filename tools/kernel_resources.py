#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of the device library, from hipcc's own
`-Rpass-analysis=kernel-resource-usage` remarks (cross-compiles for gfx950: no GPU needed).

    python tools/kernel_resources.py [extra hipcc flags ...]      e.g.  -DRT_SHADE_WAVES=3
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "rust_raytracer_amd", "csrc", "rt_kernels.hip")
CXXFILT = "c++filt"


def main():
    extra = sys.argv[1:]
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
               "--offload-device-only", "-c", "-Rpass-analysis=kernel-resource-usage", SRC,
               "-o", os.path.join(tmp, "k.o")] + extra
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr)
            raise SystemExit(r.returncode)
    blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
    names = [b.split()[0] for b in blocks]
    dem = subprocess.run([CXXFILT] + names, capture_output=True, text=True).stdout.splitlines()
    keys = [("VGPR", r"VGPRs: (\d+)"), ("AGPR", r"AGPRs: (\d+)"), ("SGPR", r"SGPRs: (\d+)"),
            ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("waves/SIMD", r"Occupancy \[waves/SIMD\]: (\d+)"),
            ("LDS", r"LDS Size \[bytes/block\]: (\d+)")]
    print(f"{'kernel':78s} " + " ".join(f"{k:>10s}" for k, _ in keys))
    for b, name in zip(blocks, dem):
        name = name.replace("rt::", "").replace("void ", "")
        name = re.sub(r"\(.*", "", name)
        vals = []
        for _, pat in keys:
            m = re.search(pat, b)
            vals.append(m.group(1) if m else "?")
        print(f"{name[:78]:78s} " + " ".join(f"{v:>10s}" for v in vals))


if __name__ == "__main__":
    main()
