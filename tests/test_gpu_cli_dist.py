"""End-to-end GPU tests of the things around the kernels: the `rtrace` executable (drop-in for the reference
binary, src/main.rs:25-88), the multi-rank render + gather with the HIP renderer on every rank, and bench.py's own
N-rank launch.  All of them need the MI355X box; multi-rank cases put their (2-3) ranks on the one GPU there."""
import json
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import pyoracle
from rust_raytracer_amd import api

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RTRACE = os.path.join(REPO, "rust_raytracer_amd", "rtrace")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_rtrace(args, cwd, env=None):
    r = subprocess.run([RTRACE] + args, cwd=cwd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_rtrace_binary_end_to_end(tmp_path):
    """flags -> scene DSL -> rt_render -> ACES / sRGB -> out.png, plus the console lines of main.rs:61-85."""
    assert os.path.exists(RTRACE)
    args = [os.path.join(REPO, "scenes", "cornell"), "-w=64", "-s=16", "--seed=1"]
    out = run_rtrace(args, str(tmp_path))
    lines = out.strip().splitlines()
    dur = r"\d+\.\d\d(ns|µs|ms|s)"
    assert re.fullmatch(rf"Ready: {dur}", lines[0])                                              # main.rs:62
    assert lines[1] == "Rendering: 64x64 @16spp on 1 threads (16 samples/thread)"                # main.rs:68-71
    assert re.fullmatch(rf"GPU 0 finished in {dur}", lines[2])                                   # camera.rs:236 (per thread there)
    assert re.fullmatch(rf"Done: {dur}\. Writing output to file\.\.\.", lines[3])                # main.rs:78
    assert re.fullmatch(rf"Done! Took {dur}\. Goodbye :\)", lines[4])                            # main.rs:85
    assert len(lines) == 5
    png = api.load_image(str(tmp_path / "out.png"))            # (h, w, 3) float32 = byte / 255
    got = np.rint(png * 255.0).astype(np.uint8)
    hs = api.HostScene(["scenes/cornell", "-w=64", "-s=16", "--seed=1"])
    gpu = api.DeviceScene(hs.desc, 0).render(hs.camera, hs.params)
    np.testing.assert_array_equal(got, api.tonemap_rgb8(gpu))   # the file holds exactly the output stage of the frame
    ref, _ = pyoracle.render(hs.desc, hs.camera, hs.params)
    want = api.tonemap_rgb8(ref)                                # reference output stage on the oracle's frame
    diff = np.abs(got.astype(int) - want.astype(int))
    assert diff.max() <= 1 and (diff != 0).mean() < 1e-3        # 1e-15 relative differences can only move a value on a rounding edge
    # --gpus=1 is the default; a mesh scene with replicas goes through the same path
    (tmp_path / "b").mkdir()
    out_b = run_rtrace(args + ["--gpus=1"], str(tmp_path / "b"))
    assert out_b.splitlines()[1] == lines[1]
    assert (tmp_path / "b" / "out.png").read_bytes() == (tmp_path / "out.png").read_bytes()
    (tmp_path / "c").mkdir()
    out_c = run_rtrace([os.path.join(REPO, "scenes", "light_test"), "-w=60", "-s=18", "-t=2", "--seed=2"], str(tmp_path / "c"))
    assert out_c.splitlines()[0] == "Loaded 15744 tris"                                          # loaders/obj.rs
    assert "Rendering: 60x40 @18spp on 2 threads (9 samples/thread)" in out_c
    # --gpus=3 (rehearsed on this box's one GPU: RT_RTRACE_ONE_DEVICE=1 puts every part on device 0): three host threads, three
    # row parts in rth_band_rows bands, assembled on the host: the same out.png, byte for byte, as --gpus=1
    for sub, extra in (("d1", ["--gpus=1"]), ("d3", ["--gpus=3"])):
        (tmp_path / sub).mkdir()
        out_d = run_rtrace([os.path.join(REPO, "scenes", "light_test"), "-w=90", "-s=18", "-t=2", "--seed=2"] + extra, str(tmp_path / sub),
                           env=dict(os.environ, RT_RTRACE_ONE_DEVICE="1"))
        assert sum(1 for ln in out_d.splitlines() if re.fullmatch(rf"GPU \d finished in {dur}", ln)) == (3 if sub == "d3" else 1)
    assert (tmp_path / "d3" / "out.png").read_bytes() == (tmp_path / "d1" / "out.png").read_bytes()
    # bad input: message on stderr, non-zero exit, no abort
    r = subprocess.run([RTRACE, "/nonexistent/scene"], cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 1 and r.stderr.startswith("Error:")


@pytest.mark.parametrize("world", [2, 3])
def test_row_tiled_render_and_gather_with_the_hip_renderer(tmp_path, world):
    """dist.render_distributed with rt_render_device on every rank (16-row bands, padded gather, de-interleave):
    the assembled frame is the single-GPU frame bit for bit."""
    out = tmp_path / "result.txt"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(REPO, "tests", "dist_worker_gpu.py"), str(out)]
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert out.read_text() == "OK"


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it must start two ranks itself (from a parent that has made
    no GPU call) and report n_gpus = 2; a launcher whose world size disagrees with --gpus is an error."""
    env = dict(os.environ, RT_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")   # both ranks on cuda:0, gloo
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--workload", "c2",
           "--spp-divisor", "16", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 4 and res["value"] > 0 and res["scaling"] == "strong"
    # every N renders its timed frames one after the other by default (the N = 1 and N > 1 lines then measure the same thing) ...
    assert res["config"]["frames_in_flight"] == 1 and res["value_basis"].startswith("single frames")
    assert "REHEARSAL" in res["config"]["partition"]
    # ... and pipelining the frames of a sequence (api.FramePipeline) is opt-in and labelled
    r3 = subprocess.run(cmd + ["--frames-in-flight", "3"], cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r3.returncode == 0, r3.stdout[-3000:] + r3.stderr[-3000:]
    res3 = json.loads([ln for ln in r3.stdout.splitlines() if ln.startswith("{")][0])
    assert res3["n_gpus"] == 2 and res3["config"]["frames_in_flight"] == 3 and res3["value_basis"].startswith("pipelined")
    bad = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "3", "--steps", "1", "--no-cpu-baseline"],
                         cwd=REPO, env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "WORLD_SIZE=2" in (bad.stdout + bad.stderr)
