#!/usr/bin/env python3
"""bench.py — Msamples/s of the render hot path on N MI355X GPUs of one node.

A "step" is ONE full pass of the hot path (`rt_render_device`, the drop-in for the
reference's `Camera::render`, src/camera.rs:189) over the headline workload
BASELINE.json names: scenes/cornell_dragon, 1200x1200, 1000 spp (10 replicas x 10x10
strata), 870k-triangle mesh (deterministic stand-in, see tools/gen_dragon.cpp).  The scene
(BVH, tables) is resident in HBM before the timed region; the frame stays in HBM.

N > 1 (launched by torch.distributed.run, one rank per GPU): the SAME frame is row-tiled in
interleaved 16-row bands across the ranks, each rank renders all samples of its rows, and
one RCCL gather per step assembles the frame on rank 0 (inside the timed region).  Total
work is fixed as N grows: "scaling": "strong".

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and
`cpu_baseline` objects.  The CPU baseline runs the oracle (oracle/liboracle.so, a
structure-faithful restatement of the reference's CPU path) on a bounded row subset.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec

WORKLOADS = {
    # name: (scene args, description)
    "c4": (["scenes/cornell_dragon", "-w=1200", "-s=1000", "-t=10"],
           "scenes/cornell_dragon 1200x1200 @1000spp (10 replicas x 10x10 strata), 871200-tri stand-in mesh"),
    "c3": (["scenes/light_test", "-w=1200", "-s=1000", "-t=10"],
           "scenes/light_test 1200x800 @1000spp, Suzanne 15.7k tri"),
    "c2": (["scenes/cornell", "-w=800", "-s=256"], "scenes/cornell 800x800 @256spp, quads + glass sphere"),
    # the reference's built-in default scene (main.rs without a scene file: 440-sphere field in its object-BVH, Suzanne,
    # sun + sky) at the size of samples/sample0.png; BASELINE config C1 is this scene at 400x266 @64spp on the CPU
    "c1": (["-w=1200", "-s=256", "-t=4"], "default scene (golden_monkey.rs) 1200x800 @256spp, 440 spheres in an object BVH + Suzanne 15.7k tri"),
    # rows f-3 / f-4 of SURVEY section 8: more than one mesh instance in `world`, and constant-density volumes
    # the headline scene with a finer tessellation of the same stand-in surface: BVH nodes + triangle records (the traversal
    # kernel's hot set) are 314 MB / 1.25 GB instead of 78 MB, i.e. beyond the 256 MB Infinity Cache (MI355X_MICROARCH.md:
    # "scale past L3 before reading FETCH_SIZE"); scene file and OBJ are generated under build/bigmesh/ on first use
    "c4_3m": (["build/bigmesh/cornell_dragon_3m", "-w=1200", "-s=1000", "-t=10"],
              "scenes/cornell_dragon 1200x1200 @1000spp with the stand-in mesh at 1320x1320 quads = 3484800 triangles"),
    "c4_14m": (["build/bigmesh/cornell_dragon_14m", "-w=1200", "-s=1000", "-t=10"],
               "scenes/cornell_dragon 1200x1200 @1000spp with the stand-in mesh at 2640x2640 quads = 13939200 triangles"),
    # the other direction (does k_wf_mesh get faster when its working set fits the L2s?): 330 / 165 quads per side
    "c4_218k": (["build/bigmesh/cornell_dragon_218k", "-w=1200", "-s=1000", "-t=10"],
                "scenes/cornell_dragon 1200x1200 @1000spp with the stand-in mesh at 330x330 quads = 217800 triangles"),
    "c4_54k": (["build/bigmesh/cornell_dragon_54k", "-w=1200", "-s=1000", "-t=10"],
               "scenes/cornell_dragon 1200x1200 @1000spp with the stand-in mesh at 165x165 quads = 54450 triangles"),
    "two_meshes": (["tests/scenes/two_meshes", "-w=800", "-s=256"], "tests/scenes/two_meshes 800x800 @256spp, two transformed mesh instances (2 x 967 tri)"),
    "smoke": (["scenes/cornell_smoke", "-w=800", "-s=256"], "scenes/cornell_smoke 800x800 @256spp, two constant-density volumes bounded by boxes"),
}


def ensure_dragon():
    path = os.path.join(REPO, "scenes", "resource", "dragon_high.obj")
    if not os.path.exists(path):
        tool = os.path.join(REPO, "tools", "gen_dragon")
        if not os.path.exists(tool):
            subprocess.run(["g++", "-std=c++17", "-O2", "-o", tool, tool + ".cpp"], check=True)
        tmp = path + f".tmp{os.getpid()}"
        subprocess.run([tool, tmp], check=True)
        os.replace(tmp, path)


BIG_MESHES = {"c4_3m": ("3m", 1320), "c4_14m": ("14m", 2640), "c4_218k": ("218k", 330), "c4_54k": ("54k", 165)}


def ensure_big_dragon(workload: str):
    """build/bigmesh/cornell_dragon_<tag> + dragon_<tag>.obj: scenes/cornell_dragon with tools/gen_dragon's surface at NU = NV = n."""
    tag, n = BIG_MESHES[workload]
    d = os.path.join(REPO, "build", "bigmesh")
    os.makedirs(d, exist_ok=True)
    obj = os.path.join(d, f"dragon_{tag}.obj")
    if not os.path.exists(obj):
        tool = os.path.join(REPO, "tools", "gen_dragon")
        if not os.path.exists(tool):
            subprocess.run(["g++", "-std=c++17", "-O2", "-o", tool, tool + ".cpp"], check=True)
        tmp = obj + f".tmp{os.getpid()}"
        subprocess.run([tool, tmp, str(n), str(n)], check=True)
        os.replace(tmp, obj)
    scene = os.path.join(d, f"cornell_dragon_{tag}")
    text = open(os.path.join(REPO, "scenes", "cornell_dragon")).read().replace("resource/dragon_high.obj", f"dragon_{tag}.obj")
    with open(scene + ".tmp", "w") as f:
        f.write(text)
    os.replace(scene + ".tmp", scene)


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def physical_cores():
    """One hardware thread per physical core of the CPUs this process may run on (SMT siblings dropped), grouped by socket:
    returns the cores of the socket that offers most of them (the oracle's scene then lives in that socket's memory: the
    process is pinned there BEFORE the scene is loaded, first touch)."""
    allowed = sorted(os.sched_getaffinity(0))
    seen, by_pkg = set(), {}
    for cpu in allowed:
        base = f"/sys/devices/system/cpu/cpu{cpu}/topology/"
        try:
            with open(base + "thread_siblings_list") as f:
                sib = f.read().strip()
            with open(base + "physical_package_id") as f:
                pkg = f.read().strip()
        except OSError:
            sib, pkg = str(cpu), "0"
        if (pkg, sib) not in seen:
            seen.add((pkg, sib))
            by_pkg.setdefault(pkg, []).append(cpu)
    return max(by_pkg.values(), key=len)


def _oracle_rows(hs, p, stride):
    """The oracle on an unbiased ROW subset (every `stride`-th row), as two disjoint interleaved halves timed separately."""
    from oracle import pyoracle

    halves = []
    tot = {"samples": 0, "seconds": 0.0, "rays": 0, "nodes": 0, "tris": 0}
    for part in (stride // 4, stride // 4 + stride // 2):
        q = p.copy()
        q.band_rows, q.n_parts, q.part = 1, stride, part
        _, st = pyoracle.render(hs.desc, hs.camera, q)
        halves.append(st.samples / st.seconds / 1e6)
        tot["samples"] += st.samples; tot["seconds"] += st.seconds
        tot["rays"] += st.rays; tot["nodes"] += st.node_tests; tot["tris"] += st.tri_tests
    return halves, tot


def cpu_baseline(workload: str, seed: int):
    """Times the CPU oracle on a bounded sample of the same workload (rank 0, N=1 only).

    Exactly the workload's flags (same -t replicas = OS threads like the reference's -t, same s x s
    strata, so every pixel gets the workload's spp), on an unbiased ROW subset of the frame (every
    `stride`-th row, not a crop — BASELINE.md section 3 plans 1/16 of the rows, which is > 60 s of CPU work at
    C4; the sample here is sized for ~20 s).  The subset is rendered as TWO disjoint interleaved halves
    that are timed separately: their spread is the stated variance of the estimate.

    `all_cores` (workloads whose spp allows it): the same rows with the samples spread over as many replicas (= OS
    threads, src/config.rs:154-155: spp = T * floor(sqrt(s / T))^2) as the host has physical cores for, at the SAME spp
    (-t=40: 40 x 5 x 5 = 1000), pinned to one hardware thread per physical core: what this CPU can really do."""
    from rust_raytracer_amd import api

    args, _ = WORKLOADS[workload]
    avail = len(os.sched_getaffinity(0))
    saved_affinity = os.sched_getaffinity(0)
    cores = physical_cores()
    os.sched_setaffinity(0, set(cores))  # one socket, one hardware thread per core; the scene is loaded (first touch) under it
    try:
        return _cpu_baseline_pinned(api, args, seed, avail, cores)
    finally:
        os.sched_setaffinity(0, saved_affinity)


def _cpu_baseline_pinned(api, args, seed, avail, cores):
    hs = api.HostScene(args + [f"--seed={seed}"])
    p = hs.params.copy()
    threads = p.thread_count  # one OS thread per replica (camera.rs:197-241)
    target_samples = 1.3e6 * min(threads, avail) * 0.9  # ~25 s at the oracle's speed on this class of host
    per_row = hs.width * p.sqrt_spt * p.sqrt_spt * threads
    n_rows = max(2, int(round(target_samples / per_row)))
    stride = max(2, hs.height // n_rows)
    halves, tot = _oracle_rows(hs, p, stride)
    value = tot["samples"] / tot["seconds"] / 1e6
    rows = tot["samples"] // per_row
    out = {
        "value": value,
        "unit": "Msamples/s",
        "cores": min(threads, avail),
        "kind": "port",
        "sample": f"{rows} of {hs.height} rows (every {stride}th, two interleaved halves) x {hs.width} px x {threads} replicas "
                  f"(= OS threads, the workload's -t) x {p.sqrt_spt * p.sqrt_spt} strata = {tot['samples']} samples in "
                  f"{tot['seconds']:.1f}s (octree build excluded); halves {halves[0]:.3f} / {halves[1]:.3f} Msamples/s "
                  f"(spread {abs(halves[0] - halves[1]) / value:.1%}); oracle/liboracle.so -O3 x86-64-v3, f64, recursive, "
                  f"reference octree; keyed per-sample RNG (the reference's is one sequential stream per thread)",
        "cpu_model": cpu_model(),
        "host_cores_available": avail,
        "halves": halves,
        "rays_per_sample": tot["rays"] / max(tot["samples"], 1),
        "node_tests_per_ray": tot["nodes"] / max(tot["rays"], 1),
        "tri_tests_per_ray": tot["tris"] / max(tot["rays"], 1),
    }
    out["pinned_to"] = f"{len(cores)} physical cores of one socket (one hardware thread each)"
    # the same spp on more replicas: T' = the largest T' <= physical cores with T' * floor(sqrt(s / T'))^2 == spp
    spp = threads * p.sqrt_spt * p.sqrt_spt
    best = None
    for t2 in range(min(len(cores), 128), threads, -1):
        s2 = int((spp // t2) ** 0.5)
        if s2 >= 1 and t2 * s2 * s2 == spp:
            best = (t2, s2)
            break
    if best:
        t2, s2 = best
        q = p.copy()
        q.thread_count, q.sqrt_spt = t2, s2
        halves2, tot2 = _oracle_rows(hs, q, stride)
        v2 = tot2["samples"] / tot2["seconds"] / 1e6
        out["all_cores"] = {
            "value": v2, "unit": "Msamples/s", "cores": t2, "physical_cores_available": len(cores),
            "sample": f"the same {rows} rows at the same spp as -t={t2} x {s2}x{s2} strata ({t2} OS threads on the {len(cores)} physical cores of one "
                      f"socket, one hardware thread per core): {tot2['samples']} samples in {tot2['seconds']:.1f}s; halves {halves2[0]:.3f} / {halves2[1]:.3f}",
            "halves": halves2,
        }
    return out


# Second ceiling of these kernels: wave-level VALU instructions per second.  A wave64 instruction occupies a SIMD-32 for 4 clk
# (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost'): 1024 SIMDs x 2.4 GHz / 4 = 614.4 G instructions/s.  f64 divisions,
# square roots and reciprocals issue at a quarter of that rate, so the share computed from the instruction COUNT is a lower bound.
VALU_PEAK_GINST = 1024 * 2.4 / 4.0


def profiled_entry(workload: str, precision: str, launches: float):
    """Counter figures of this workload from profiles/traffic.json (tools/make_traffic.py), or (None, reason) when the workload
    was not profiled or the profile is stale: taken on other library sources, or at another number of launches per step."""
    try:
        entry = json.load(open(os.path.join(REPO, "profiles", "traffic.json"))).get("entries", {}).get(f"{workload}_{precision}")
    except Exception:
        entry = None
    if not entry:
        return None, "this workload has no rocprofv3 counter profile under profiles/"
    try:
        digest = open(os.path.join(REPO, "rust_raytracer_amd", "librt_mi355.so.srchash")).read().strip()
    except OSError:
        digest = ""
    if entry.get("lib_digest") != digest:
        return None, f"stale profile ({entry.get('profile')}): the library sources have changed since it was taken"
    if abs(float(entry.get("launches_per_step", -1)) - launches) > 0.5:
        return None, f"stale profile ({entry.get('profile')}): {entry.get('launches_per_step')} launches per step then, {launches} now"
    return entry, None


def make_roofline(api, counters, kstats, a, owned_pixels, ms_per_step):
    """`roofline` object for the DOMINANT kernel of this workload (largest HIP-event sum per step).

    Algorithmic bytes are counted by the kernels themselves in the counter-collecting warm-up step (deterministic
    for a given seed / config) and divided by the launches per step; durations are HIP-event times on the render
    stream, summed per kernel by the library (RtRenderStats).  No kernel here is a dense contraction (no MFMA); each is
    priced against BOTH ceilings it can meet and `bound` names the one it is closer to:
      hbm    bytes / kernel time / 8 TB/s.  Algorithmic bytes:
               k_wf_mesh / k_wf_intersect / k_megakernel: BVH nodes fetched x node size + triangle tests x record size
                                                          (+ path state of the rays handled)
               k_wf_shade / k_wf_prims:                   path-state bytes read + written per ray
             and, when this workload has a current counter profile, the fabric-side bytes rocprofv3 measured (`traffic`).
             A kernel whose algorithmic rate exceeds the HBM peak is running out of L2 / Infinity Cache: its fraction is
             then the measured one, never > 1.
      valu   wave-level VALU instructions (SQ_INSTS_VALU of the profile) / kernel time / 614.4 G/s: the f64 arithmetic
             of the path tracer (a lower bound of the issue time: quarter-rate f64 division / sqrt steps count once)."""
    n = len(kstats)
    mean = lambda k: sum(x[k] for x in kstats) / n
    launches = mean("launches")
    mega = counters.pipeline_used == api.RT_PIPELINE_MEGAKERNEL
    bvh_bytes = counters.node_visits * counters.bytes_node + counters.tri_tests * counters.bytes_tri
    if mega:
        cands = {"k_megakernel": (mean("all"), bvh_bytes + counters.mesh_rays * counters.bytes_attr + owned_pixels * 32, 1.0)}
    else:
        cands = {
            "k_wf_mesh (BVH traversal; k_wf_intersect for scenes the split kernels do not cover)":
                (mean("traversal"), bvh_bytes + counters.mesh_rays * counters.bytes_state, launches),
            "k_wf_shade (scatter, pdf, regeneration, compaction)":
                (mean("shade"), counters.rays * counters.bytes_state_shade + counters.samples * 24, launches),
            "k_wf_prims (scene program over spheres / quads / sky / sun)":
                (mean("prims"), counters.rays * counters.bytes_state_prims, launches),
        }
    name, (ms_step, nbytes_step, n_launch) = max(cands.items(), key=lambda kv: kv[1][0])
    if ms_step <= 0:
        return None
    short = name.split()[0]
    avg_ms = ms_step / n_launch
    nbytes = nbytes_step / n_launch
    alg_rate = nbytes / (avg_ms * 1e-3) / 1e9
    entry, why_not = profiled_entry(a.workload, a.precision, 1.0 if mega else launches)
    kprof = (entry or {}).get("kernels", {}).get(short) if entry else None
    traffic = kprof["bytes_per_step"] / n_launch if kprof else None
    frac_traffic = None if traffic is None else traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    valu_rate = valu_share = None
    if kprof and kprof.get("valu_insts_per_step"):
        valu_rate = kprof["valu_insts_per_step"] / (ms_step * 1e-3) / 1e9
        valu_share = valu_rate / VALU_PEAK_GINST
    cache_resident = alg_rate > HBM_PEAK_GBS
    if not cache_resident:
        hbm_frac, hbm_basis = alg_rate / HBM_PEAK_GBS, "algorithmic bytes / kernel time / HBM peak"
    else:
        hbm_frac = frac_traffic
        hbm_basis = ("counter bytes (2 x FETCH_SIZE + WRITE_SIZE, profiles/traffic.json) / kernel time / HBM peak: the algorithmic rate "
                     f"({alg_rate:.0f} GB/s) exceeds the HBM peak, L1 / L2 / Infinity Cache serve part of it")
    valu_bound = valu_share is not None and (hbm_frac is None or valu_share > hbm_frac)
    out = {
        "bound": "valu" if valu_bound else "hbm",
        "achieved": valu_rate if valu_bound else alg_rate,
        "peak": VALU_PEAK_GINST if valu_bound else HBM_PEAK_GBS,
        "unit": "G wave-instructions/s" if valu_bound else "GB/s",
        "frac": valu_share if valu_bound else hbm_frac,
        "frac_basis": ("VALU instructions of the profiled run (SQ_INSTS_VALU, profiles/traffic.json) / this run's kernel time / (1024 SIMDs x 2.4 GHz / 4 clk): "
                       "above this kernel's HBM fraction, so vector issue is the nearer ceiling") if valu_bound else hbm_basis,
        "traffic": traffic,
        "hbm": {"achieved": alg_rate, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_frac, "frac_basis": hbm_basis,
                "traffic": traffic, "frac_traffic": frac_traffic},
        "valu": {"achieved": valu_rate, "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s", "frac": valu_share},
        "profile": (entry or {}).get("profile") if kprof else None,
        "profile_note": why_not if not kprof else None,
        "kernel": name,
        "note": ("HIP-event kernel time on the render stream; algorithmic bytes counted by the kernels.  `traffic` = rocprofv3 counter bytes of the same "
                 "kernel per launch (2 x FETCH_SIZE + WRITE_SIZE, fabric side: Infinity-Cache hits are counted; calibration in "
                 "profiles/r01/fetch_size_calibration.txt), present only while profiles/traffic.json holds a profile of THIS library build at THIS "
                 "number of launches per step (`profile_note` says why not)."),
        "dominant_kernel_share_of_step": ms_step / ms_per_step,
        "kernel_ms_avg": avg_ms, "launches_per_step": n_launch, "algorithmic_bytes_per_launch": nbytes,
        "kernel_ms_per_step": ms_step,
        "all_kernels_ms_per_step": {k.split()[0]: v[0] for k, v in cands.items()},
        "rays_per_sample": counters.rays / max(counters.samples, 1),
        "mesh_rays_per_ray": counters.mesh_rays / counters.rays,
        "node_visits_per_ray": counters.node_visits / counters.rays,
        "tri_tests_per_ray": counters.tri_tests / counters.rays,
        "bytes_node": counters.bytes_node, "bytes_tri": counters.bytes_tri,
        "bytes_state_shade": counters.bytes_state_shade, "bytes_state_prims": counters.bytes_state_prims,
    }
    if entry and not mega:  # every profiled kernel of this workload against both ceilings
        per = {}
        for k, (ms_k, nb_k, nl_k) in cands.items():
            kp = entry["kernels"].get(k.split()[0])
            if not kp or ms_k <= 0:
                continue
            per[k.split()[0]] = {"ms_per_step": ms_k, "hbm_frac_algorithmic": nb_k / (ms_k * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "hbm_frac_traffic": kp["bytes_per_step"] / (ms_k * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "write_over_algorithmic": None,
                                 "valu_issue_share": (kp["valu_insts_per_step"] / (ms_k * 1e-3) / 1e9 / VALU_PEAK_GINST) if kp.get("valu_insts_per_step") else None}
        out["kernels"] = per
    if short == "k_wf_mesh" and counters.node_visits:
        # the memory system's own limit for this access pattern (dependent fetches of random 128-B lines):
        # tools/ubench/gather_lines on the same chip, profiles/r01/ubench_gather_lines.txt
        out["line_requests_per_s"] = (counters.node_visits * -(-counters.bytes_node // 128) + counters.tri_tests * counters.bytes_tri / 128.0) / n_launch / (avg_ms * 1e-3)
        out["random_line_ceiling_per_s"] = [59e9, 79e9]
    return out


def free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) with torch.distributed.run as a
    CHILD process.  This parent has made no GPU call (torch is not even imported yet), it only relays the
    child's output and checks that the world that rendered is the one that was asked for."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=REPO)
    seen = None
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
        if line.startswith("{"):
            try:
                seen = json.loads(line)
            except ValueError:
                pass
    rc = proc.wait()
    if rc != 0:
        return rc
    if seen is None or seen.get("n_gpus") != n:
        sys.stderr.write(f"bench.py: asked for {n} GPUs but the result line says {None if seen is None else seen.get('n_gpus')}\n")
        return 3
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--precision", default="f64", choices=["f64", "f32"])
    ap.add_argument("--pipeline", default="auto", choices=["auto", "mega", "wavefront"])
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--spp-divisor", type=int, default=1, help="debug: render spp/divisor (result is then labelled reduced)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="device scenes of the frame pipeline (api.FramePipeline): 1 (default, every N) = the timed frames strictly one after "
                         "the other, so that the N = 1 and N > 1 lines measure the same thing; 3 = frame k+1 starts under the tail of frame k "
                         "(throughput of a frame SEQUENCE: the line then says value_basis = pipelined)")
    ap.add_argument("--no-overlap", action="store_true", help="same as --frames-in-flight 1")
    ap.add_argument("--save-png", default="")
    a = ap.parse_args()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(launch_ranks(a.gpus))

    import torch
    import torch.distributed as dist

    from rust_raytracer_amd import api
    from rust_raytracer_amd import dist as rtdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal of the multi-rank path on a one-GPU box: RT_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and uses gloo
    # (RCCL refuses two ranks on one device).  Never set by the driver; the result is then labelled in `config`.
    one_device = os.environ.get("RT_BENCH_ONE_DEVICE", "0") == "1"
    if one_device:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP render path has no CPU fallback)")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    if a.workload == "c4" or a.workload in BIG_MESHES:
        if rank == 0:
            ensure_dragon() if a.workload == "c4" else ensure_big_dragon(a.workload)
        if world > 1:
            dist.barrier()
    args, desc = WORKLOADS[a.workload]
    args = list(args)
    if a.spp_divisor > 1:
        s = int([x for x in args if x.startswith("-s=")][0][3:]) // a.spp_divisor
        args = [x for x in args if not x.startswith("-s=")] + [f"-s={s}"]
    hs = api.HostScene(args + [f"--seed={a.seed}", f"--precision={a.precision}", f"--pipeline={a.pipeline}"])
    # The K timed steps are K frames of the same job, rendered strictly one after the other at every N (depth 1), so that the
    # N = 1 and N > 1 lines measure the same thing.  --frames-in-flight 3 is the opt-in throughput mode for frame SEQUENCES on a
    # multi-GPU rank (api.FramePipeline: frame k+1 starts under the tail of frame k, rt_scene_set_tail_flag; measured on a 1/8
    # share of the headline frame: 173 -> 166 ms per frame, tools/gpu_pipeline_probe.py); the line then says so in `value_basis`.
    depth = a.frames_in_flight if a.frames_in_flight > 0 else 1
    if a.no_overlap or a.steps < 2:
        depth = 1
    pipe = api.FramePipeline(hs.desc, local_rank, depth)  # BVH build + upload: resident before timing
    scene = pipe.scenes[0]
    params = rtdist.partition_params(hs.params, world, rank, hs.height)
    rows = len(rtdist.rows_of_part(hs.height, world, rank))
    n_warm = max(a.warmup, depth)  # every device scene of the pipeline renders at least one untimed frame (pool allocation)
    n_buf = max(a.steps, n_warm)
    outs = [torch.empty((rows, hs.width, 4), dtype=torch.float64, device=device) for _ in range(n_buf)]
    streams = [torch.cuda.Stream(device) for _ in range(depth)]
    stream_handles = [st.cuda_stream for st in streams]

    frame_holder = {}

    def run(n_frames, collect_first=False):
        plist = []
        for k in range(n_frames):
            p = params.copy()
            p.collect_stats = 1 if (collect_first and k == 0) else 0
            plist.append(p)
        stats = pipe.render_frames(hs.camera, plist, [outs[k].data_ptr() for k in range(n_frames)], stream_handles)
        for k in range(n_frames):  # the renders have returned (stream-synchronised by the library): one gather per frame
            frame_holder["frame"] = rtdist.gather_frame(outs[k], hs.height, hs.width)
        return stats

    # Warmup; the first warmup step also collects the traversal counters (deterministic for a
    # given seed/config) that turn kernel time into algorithmic bytes.
    counters = run(n_warm, collect_first=True)[0]  # --warmup 0 still gets the untimed counter-collecting step(s)
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    # per timed step: HIP-event sums per kernel (library side, on the render's stream) + launch counts
    kstats = [{"traversal": st_.traversal_kernel_ms, "prims": st_.prims_kernel_ms, "shade": st_.shade_kernel_ms,
               "all": st_.kernel_ms, "launches": max(st_.n_launches, 1)} for st_ in run(a.steps)]
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if one_device else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_samples = hs.width * hs.height * hs.spp  # whole job, all ranks
    value = total_samples * a.steps / elapsed / 1e6

    roofline = None
    if counters is not None and counters.rays > 0 and kstats:
        roofline = make_roofline(api, counters, kstats, a, rows * hs.width, elapsed / a.steps * 1e3)
        if roofline and depth > 1:
            roofline["note"] += (f"  {depth} frames are in flight on this rank: a kernel's HIP-event duration includes the time it shares the "
                                 "GPU with the neighbouring frames' tail launches; the N = 1 line has the undisturbed figures.")

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(a.workload, a.seed)

    if rank == 0:
        if a.save_png and frame_holder.get("frame") is not None:
            api.save_png(a.save_png, frame_holder["frame"].cpu().numpy())
        line = {
            "metric": "Msamples/sec (whole node), cornell_dragon 870k tri @1000spp" if a.workload == "c4"
                      else f"Msamples/sec (whole node), {a.workload}",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
            "config": {"workload": desc + (f" [DEBUG spp/{a.spp_divisor}]" if a.spp_divisor > 1 else ""),
                       "image": [hs.width, hs.height], "spp": hs.spp, "seed": a.seed,
                       "pipeline": a.pipeline, "frames_in_flight": depth, "untimed_frames": n_warm, "partition": f"{params.band_rows or hs.height}-row bands round-robin over {world} GPU(s)"
                                    + (" [REHEARSAL: all ranks on one device, gloo]" if one_device else "")},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        line["value_basis"] = "single frames, one after the other" if depth == 1 else f"pipelined sequence, {depth} frames in flight"
        if cpu:
            line["speedup_vs_cpu_baseline"] = value / cpu["value"]
            if cpu.get("all_cores"):
                line["speedup_vs_cpu_all_cores"] = value / cpu["all_cores"]["value"]
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
