"""Multi-process path on CPU: world_size 2 and 3 over gloo (the GPU run uses the same code over
RCCL).  Checks the interleaved 16-row band partition, the padded equal-size gather and the
de-interleave on rank 0 against a single-process render."""
import os
import socket
import subprocess
import sys

import pytest

from rust_raytracer_amd import api
from rust_raytracer_amd import dist as rtdist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_partition_is_a_partition():
    for h in (1, 15, 16, 17, 80, 1200):
        for n in (1, 2, 3, 4, 8):
            rows = [rtdist.rows_of_part(h, n, r) for r in range(n)]
            flat = sorted(y for part in rows for y in part)
            assert flat == list(range(h))
            assert rtdist.max_rows(h, n) == max(len(r) for r in rows)
    # 1200 rows over 8 GPUs: 16-row bands would give 160 or 144 rows per rank; the band height is chosen so that the most
    # loaded rank (it sets the time of the run) has as few rows as possible: 2-row bands, 150 rows each
    assert rtdist.band_rows_for(1200, 8) == 2
    assert [len(rtdist.rows_of_part(1200, 8, r)) for r in range(8)] == [150] * 8
    assert rtdist.band_rows_for(2400, 8) == 4 and rtdist.band_rows_for(1200, 2) == 8 and rtdist.band_rows_for(1200, 3) == 16
    for h in (1, 15, 16, 17, 80, 266, 800, 1200, 2400):
        for n in (2, 3, 4, 8):
            b = rtdist.band_rows_for(h, n)
            most = max(len(rtdist.rows_of_part(h, n, r)) for r in range(n))
            assert all(most <= max(len(rtdist.rows_of_part_banded(h, n, r, bb)) for r in range(n)) for bb in (16, 8, 4, 2, 1)), (h, n, b)
            p = rtdist.partition_params(api.RtRenderParams(), n, 0, h)
            assert p.band_rows == b and sorted(y for r in range(n) for y in rtdist.rows_of_part(h, n, r)) == list(range(h))
            # `rtrace --gpus=N` (csrc/host/main.cpp) takes its band height from the native twin of the rule: the same choice
            assert api.load_host_lib().rth_band_rows(h, n) == b, (h, n)
    assert api.load_host_lib().rth_band_rows(1200, 1) == 0 and rtdist.band_rows_for(1200, 1) == 0


@pytest.mark.parametrize("world", [2, 3])
def test_gather_over_gloo(tmp_path, world):
    out = tmp_path / "result.txt"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(REPO, "tests", "dist_worker.py"), str(out)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert out.read_text() == "OK"
