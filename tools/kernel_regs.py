#!/usr/bin/env python3
"""VGPRs, scratch bytes and LDS of every kernel in a device-only assembly dump of rt_kernels.hip:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -Iinclude -o all.s rust_raytracer_amd/csrc/rt_kernels.hip
    python tools/kernel_regs.py all.s [name-substring]
"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
names, rows = [], []
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    body = m.group(2)
    names.append(m.group(1))
    rows.append((re.search(r"next_free_vgpr (\d+)", body).group(1), re.search(r"private_segment_fixed_size (\d+)", body).group(1),
                 re.search(r"group_segment_fixed_size (\d+)", body).group(1)))
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
for d, (v, s, l) in zip(dem, rows):
    d = d.split("(")[0].replace("void rt::", "")
    if want in d:
        print(f"{d:50s} vgpr {v:>4s}  scratch {s:>4s} B  static lds {l} B")
