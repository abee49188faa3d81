"""C5 (cornell_dragon 2400x2400 @4000 spp = 10 replicas x 20x20 strata) as ONE rank of an 8-GPU run sees it:
part 0 of 8 interleaved 4-row bands (dist.band_rows_for: 300 rows, 2.88 G samples; the per-sample buffer needs replica groups).
Checks three of those rows against a render that owns only them, and prints the rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rust_raytracer_amd import api
from rust_raytracer_amd import dist as rtdist
import bench
bench.ensure_dragon()
hs = api.HostScene(["scenes/cornell_dragon", "-w=2400", "-s=4000", "-t=10", "--seed=1"])
assert hs.spp == 4000 and hs.height == 2400
scene = api.DeviceScene(hs.desc, 0)
p = rtdist.partition_params(hs.params, 8, 0, hs.height)
rows = rtdist.rows_of_part(hs.height, 8, 0)
dev = torch.device("cuda", 0)
out = torch.empty((len(rows), hs.width, 4), dtype=torch.float64, device=dev)
st = torch.cuda.current_stream(dev)
t0 = time.perf_counter()
scene.render_device(hs.camera, p, out.data_ptr(), st.cuda_stream)
torch.cuda.synchronize()
t = time.perf_counter() - t0
samples = len(rows) * hs.width * hs.spp
print(f"C5 part 0/8: {len(rows)} rows, {samples/1e9:.2f} G samples in {t:.2f} s = {samples/t/1e6:.0f} Msamples/s per GPU", flush=True)
# spot check: row 128 (4-row band 32 -> part 0) rendered alone
q = hs.params.copy()
q.band_rows, q.n_parts, q.part = 1, 2400, 128
alone = scene.render(hs.camera, q)
idx = rows.index(128)
same = np.array_equal(out[idx].cpu().numpy(), alone[0])
print("row 128 of the partitioned render == the row rendered alone:", same, flush=True)
assert same
