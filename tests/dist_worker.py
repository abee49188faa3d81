"""Worker for tests/test_dist_gloo.py: exercises rust_raytracer_amd.dist (row partition + gather
+ de-interleave) with world_size > 1 on CPU (gloo).  The per-rank renderer here is the CPU oracle
(test infrastructure); on GPUs bench.py plugs rt_render_device into the same functions."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import pyoracle  # noqa: E402
from rust_raytracer_amd import api  # noqa: E402
from rust_raytracer_amd import dist as rtdist  # noqa: E402


def main():
    out_path = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    hs = api.HostScene(["tests/scenes/single_light", "-w=40", "-r=0.5", "-s=4", "--seed=21"])   # 40 x 80: 5 bands of 16 rows

    def render_rows(p):
        assert (p.n_parts, p.part, p.band_rows) == (world, rank, rtdist.band_rows_for(hs.height, world))
        img, _ = pyoracle.render(hs.desc, hs.camera, p)
        assert img.shape[0] == len(rtdist.rows_of_part(hs.height, world, rank))
        return torch.from_numpy(img)

    frame = rtdist.render_distributed(render_rows, hs.camera, hs.params)
    if rank == 0:
        full, _ = pyoracle.render(hs.desc, hs.camera, hs.params)
        ok = frame is not None and tuple(frame.shape) == full.shape and np.array_equal(frame.numpy(), full)
        with open(out_path, "w") as f:
            f.write("OK" if ok else "MISMATCH")
    else:
        assert frame is None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
