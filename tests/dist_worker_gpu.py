"""Worker for tests/test_gpu_cli_dist.py: the multi-rank path of rust_raytracer_amd.dist with the HIP renderer
on every rank — what bench.py does per step.  On the one-GPU box all ranks share cuda:0 and the exchange runs
over gloo (RCCL refuses two ranks on one device); on a multi-GPU node the same code runs with backend nccl and
one device per rank (RT_DIST_BACKEND=nccl)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from rust_raytracer_amd import api  # noqa: E402
from rust_raytracer_amd import dist as rtdist  # noqa: E402


def main():
    out_path = sys.argv[1]
    backend = os.environ.get("RT_DIST_BACKEND", "gloo")
    local = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0
    torch.cuda.set_device(local)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    hs = api.HostScene(["scenes/light_test", "-w=120", "-s=16", "-t=2", "--seed=23"])   # 120 x 80: 5 bands of 16 rows
    scene = api.DeviceScene(hs.desc, local)
    dev = torch.device("cuda", local)

    def render_rows(p):
        assert (p.n_parts, p.part, p.band_rows) == (world, rank, rtdist.band_rows_for(hs.height, world))
        rows = len(rtdist.rows_of_part(hs.height, world, rank))
        out = torch.empty((rows, hs.width, 4), dtype=torch.float64, device=dev)
        scene.render_device(hs.camera, p, out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        return out

    frame = rtdist.render_distributed(render_rows, hs.camera, hs.params)
    if rank == 0:
        full = scene.render(hs.camera, hs.params)      # the whole frame on one GPU
        got = frame.cpu().numpy()
        ok = got.shape == full.shape and np.array_equal(got, full, equal_nan=True)
        with open(out_path, "w") as f:
            f.write("OK" if ok else "MISMATCH")
    else:
        assert frame is None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
