"""Where does the end of a render go?  Per-iteration kernel times of the wavefront scheduler (RT_WF_ITER_LOG=1, RT_WF_CHECK=1:
the host looks after every iteration, so the queue length printed is the iteration's own) for a whole small frame (C2) and
for one rank's share of the headline frame on 8 GPUs; prints the time spent in iterations below a number of queued paths."""
import os, subprocess, sys, re
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os
sys.path.insert(0, %r)
import bench
from rust_raytracer_amd import api, dist
which = sys.argv[1]
if which in ("c2", "c1", "c3"):
    hs = api.HostScene(bench.WORKLOADS[which][0])
    p = hs.params.copy()
elif which == "c4":
    bench.ensure_dragon()
    hs = api.HostScene(bench.WORKLOADS["c4"][0])
    p = hs.params.copy()
else:
    bench.ensure_dragon()
    hs = api.HostScene(bench.WORKLOADS["c4"][0])
    p = dist.partition_params(hs.params, 8, 0, hs.camera.image_height)
p.pipeline = api.RT_PIPELINE_WAVEFRONT
sc = api.DeviceScene(hs.desc, 0)
sc.render(hs.camera, p)      # untimed: allocations
print("=== timed", flush=True)
sys.stderr.write("=== timed\n"); sys.stderr.flush()
sc.render(hs.camera, p)
st = sc.stats()
print("kernel_ms %%.2f iterations %%d" %% (st.kernel_ms, st.n_iterations))
''' % REPO
for which in (sys.argv[1:] or ["c2", "c4_share"]):
    env = dict(os.environ, RT_WF_ITER_LOG="1", RT_WF_CHECK="1")
    r = subprocess.run([sys.executable, "-c", CODE, which], env=env, capture_output=True, text=True)
    err = r.stderr.split("=== timed")[-1]
    rows = [(int(m.group(1)), float(m.group(2)), float(m.group(3)), float(m.group(4)))
            for m in re.finditer(r"<= (\d+) paths queued: prims ([\d.]+) ms, traversal ([\d.]+) ms, shade ([\d.]+) ms", err)]
    total = sum(a + b + c for _, a, b, c in rows)
    print(which, r.stdout.strip().splitlines()[-1], "iterations logged", len(rows), "sum of kernel times %.2f ms" % total)
    full = max(n for n, *_ in rows) if rows else 0
    for lim in (full, full // 2, 4000000, 1000000, 262144, 65536, 16384):
        sel = [(n, a, b, c) for n, a, b, c in rows if n < lim]
        print("   iterations with < %9d paths queued: %3d, %.2f ms (%.1f %%)" % (lim, len(sel), sum(a + b + c for _, a, b, c in sel), 100.0 * sum(a + b + c for _, a, b, c in sel) / max(total, 1e-9)))
    for n, a, b, c in (rows[-60:] if os.environ.get("RT_TAIL_ROWS") else []):
        print("      %9d  prims %.3f  traversal %.3f  shade %.3f" % (n, a, b, c))
