#!/usr/bin/env python3
"""Turns the rocprofv3 counter summary of one profiled bench configuration (tools/profile.sh -> pmc_summary.json) into the
per-kernel figures bench.py attaches to its roofline object (profiles/traffic.json):

    python tools/make_traffic.py <workload> <precision> <profile dir>      e.g.  c4 f64 profiles/r03/c4_f64

Per kernel of the timed step (the variants without the stats counters): fabric-side bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1 KiB
(MI355X_MICROARCH.md: FETCH_SIZE counts 128-B requests as 64 B; calibrated on this access pattern in
profiles/r01/fetch_size_calibration.txt) and wave-level VALU instructions (SQ_INSTS_VALU).  The entry records the digest of the
library sources it was profiled on and the launches per step: bench.py reports the figures only while both still match."""
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    workload, precision, pdir = sys.argv[1:4]
    summary = json.load(open(os.path.join(pdir, "pmc_summary.json")))
    bench = [json.loads(x) for x in open(os.path.join(pdir, "bench.json")) if x.startswith("{")][0]
    digest = open(os.path.join(pdir, "lib_digest.txt")).read().strip()
    real = "double" if precision == "f64" else "float"
    kernels = {}
    for group in ("fetch", "l2", "sq"):
        for key, val in summary.get(group, {}).items():
            name, counter = key.split("|")
            m = re.match(r"(k_\w+)<(\w+), (\w+)", name)
            if not m or m.group(2) != real or m.group(3) != "false":
                continue
            kernels.setdefault(m.group(1), {})[counter] = val
    entry = {"lib_digest": digest, "launches_per_step": bench["roofline"]["launches_per_step"], "profile": os.path.relpath(pdir, REPO), "kernels": {}}
    for k, c in kernels.items():
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        entry["kernels"][k] = {"bytes_per_step": (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0,
                               "read_bytes_per_step": 2.0 * c["FETCH_SIZE"] * 1024.0, "write_bytes_per_step": c["WRITE_SIZE"] * 1024.0,
                               "valu_insts_per_step": c.get("SQ_INSTS_VALU")}
    path = os.path.join(REPO, "profiles", "traffic.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    data.setdefault("entries", {})[f"{workload}_{precision}"] = entry
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(entry, indent=1))


if __name__ == "__main__":
    main()
