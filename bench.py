#!/usr/bin/env python3
"""bench.py -- the headline benchmark: Mrays/s (primary + secondary) of the MI355X backend.

A "step" is one whole frame of the workload: frame prologue + every sample pass through generate / extend / shade /
shadow / accumulate, and at N > 1 the gather of the accumulator to rank 0.

Workload (every N): BASELINE.json configs[3] -- the 9,999,392-triangle height field (seed 0x5EED0002, full replica per
GPU), 3840x2160, 64 spp, full paths (depth cap 12, two directional lights).  At N = 1 one GPU renders the whole frame; at
N > 1 the SAME frame is split by 64x64 tiles dealt round-robin to the ranks and gathered over RCCL: total work is fixed,
so the scaling is "strong".  value = rays cast by all ranks per frame / wall time per frame.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload hf10m|soup1m|cube|bunny|hf8k|...]

With --gpus N > 1 and no launcher around it (WORLD_SIZE unset) the script starts its own N ranks, one per GPU
(raytracer_amd/dist.py launch_ranks); under `python -m torch.distributed.run` it is one of the ranks.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# peaks (MI355X_MICROARCH.md "Chip-level parameters"; VALU: one wave64 instruction per 2 cycles per SIMD, measured by
# tools/ubench/valu_rate -> profiles/r02_valu_rate.txt: 1.06 ns per fma wave-instruction per SIMD)
HBM_PEAK_GBS = 8000.0
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2.0  # 1228.8 G wave64 VALU instructions per second


def cpu_workers():
    """The reference's worker count, hardware_concurrency() - 2 (ThreadPool.h:29), on the cores this job may use: the
    smallest of the affinity mask, the cgroup CPU quota and the share the harness announces (OMP_NUM_THREADS; the GPU box
    shows every core of the machine to a job that owns 16 of them).  RT_CPU_CORES overrides.  Returns (workers, cores, source)."""
    limits = {}
    if hasattr(os, "sched_getaffinity"):
        limits["affinity mask"] = len(os.sched_getaffinity(0))
    else:
        limits["cpu_count"] = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            limits["cgroup cpu.max"] = max(1, int(int(quota) / int(period)))
    except (OSError, ValueError):
        pass
    for var in ("RT_CPU_CORES", "OMP_NUM_THREADS"):
        v = os.environ.get(var, "")
        if v.isdigit() and int(v) > 0:
            if var == "RT_CPU_CORES":
                limits = {var: int(v)}
                break
            limits[var] = int(v)
    source = min(limits, key=limits.get)
    cores = limits[source]
    return max(1, cores - 2), cores, source


def workloads():
    from raytracer_amd import scenes
    return {
        "hf10m": dict(spec=scenes.hf10m_scene, width=3840, height=2160, spp=64, max_depth=12, flat_cpu=True,
                      name="height field 2236^2 quads = 9,999,392 tris seed 0x5EED0002, 3840x2160 64spp full paths (depth cap 12) [BASELINE configs[3]]"),
        "hf10m4": dict(spec=scenes.hf10m_scene, width=3840, height=2160, spp=4, max_depth=12, flat_cpu=True,
                       name="height field 2236^2 quads = 9,999,392 tris seed 0x5EED0002, 3840x2160 4spp full paths (quick variant of configs[3])"),
        "hf8k": dict(spec=scenes.hf10m_scene, width=7680, height=4320, spp=256, max_depth=12, flat_cpu=True,
                     name="height field 9,999,392 tris, 7680x4320 256spp full paths (depth cap 12) [BASELINE configs[4]]"),
        "soup1m": dict(spec=scenes.soup1m_scene, width=1920, height=1080, spp=1, max_depth=4, flat_cpu=True,
                       name="triangle soup 2^20 tris seed 0x5EED0001, 1920x1080 1spp primary+3 bounces [BASELINE configs[2]]"),
        "cube": dict(spec=scenes.cube_scene, width=1920, height=1080, spp=16, max_depth=1, flat_cpu=False,
                     name="cube (== res/models/cube.fbx, main.cpp scene) 1920x1080 16spp primary+shadow [BASELINE configs[1]]"),
        "cube12": dict(spec=scenes.cube_scene, width=1920, height=1080, spp=16, max_depth=12, flat_cpu=False,
                       name="cube 1920x1080 16spp full paths (depth cap 12)"),
        "bunny": dict(spec=scenes.bunny_scene, width=1920, height=1080, spp=16, max_depth=1, flat_cpu=True,
                      name="stanford-bunny.obj (69,451 tris) 1920x1080 16spp primary+shadow"),
    }


KERNEL_NAMES = {"extend": "k_trace<PRIMARY|CLOSEST> / k_trace_packet<PRIMARY> (extend)", "shadow": "k_trace<SHADOW> / k_trace_packet<SHADOW>", "generate": "k_generate", "shade": "k_shade",
                "accumulate": "k_accumulate"}


def kernel_sources_sha16():
    """Fingerprint of the device code the library was built from (the kernel headers, rthip.hip and the Makefile's flags):
    profiles/traffic.json is stamped with it when the counters are collected, so a replayed counter set that no longer
    belongs to the kernels being benched is called out (the GPU box has no .git to ask)."""
    import glob
    import hashlib
    h = hashlib.sha1()
    base = os.path.join(ROOT, "raytracer_amd")
    for path in sorted(glob.glob(os.path.join(base, "csrc", "*"))) + [os.path.join(base, "Makefile")]:
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


FP32_PEAK_TFLOPS = 256 * 4 * 64 * 2 * 2.4e9 / 2.0 / 1e12  # 157.3: one wave64 FMA (128 flop) per SIMD every 2 clocks


def roofline_block(workload, ktimes, wc, spp, ktimes_alone=None, scene_bytes=None):
    """Roofline of the dominant kernel family, against the bound it actually runs into.

    The traversal kernels are VALU-bound: `bound` = "valu", `achieved` = wave64 VALU instructions
    issued per second by the family's launches (SQ_INSTS_VALU per launch, from the rocprofv3 --pmc passes of this same
    command in profiles/traffic.json, divided by the launch duration measured live here), `peak` = 256 CUs x 4 SIMDs x 2.4 GHz
    / 2 clocks per instruction, `frac` = achieved / peak.  Beside it:
      fp32       SURVEY.md 8(d)'s flop figure: 27 flop per box test x 4 boxes per wide node + 57 per triangle test, over the
                 nodes / triangles the family's rays algorithmically visit (device counters) -> achieved TFLOP/s, fraction of the
                 fp32 vector peak, and VALU lane-operations issued per algorithmic flop (an FMA-only kernel would need 0.5);
      algorithmic_GBps   8(d)'s byte figure (32 B ray + 16 B hit (any-hit: 4) + 64 B per record fetched) / launch duration: mostly
                 cache hits, so it carries no fraction of the HBM peak;
      traffic / hbm_counters   HBM bytes per launch by the counters (FETCH_SIZE x 2 + WRITE_SIZE, gfx950 corrections).
    A family with no counter data, or one the counters show closer to the HBM ceiling (k_shade when it dominates), is reported
    the contract's way: bound "hbm", achieved = algorithmic GB/s."""
    n_ext = wc["extendPrimary"] + wc["extendSecondary"]
    n_paths = wc["extendPrimary"]
    alg_bytes = {
        # fused primary rays are never stored: they carry a hit record only
        # 64 B per scene record FETCHED: per ray by the per-lane launches, once per wave by the packet launches (rthip_counters)
        "extend": 16.0 * wc["extendPrimary"] + 48.0 * wc["extendSecondary"] + 64.0 * wc["recordsFetchedExtend"],
        "shadow": 36.0 * wc["shadow"] + 64.0 * wc["recordsFetchedShadow"],
        "generate": 32.0 * n_paths,
        "shade": (32.0 + 16.0 + 32.0) * n_ext,                # ray + hit read, next ray written, per shaded ray
        "accumulate": 24.0 * n_paths / spp + 17.0 * n_paths,  # accumulator read+write per pixel + radiance and liveness byte per sample
    }
    # 8(d): slab test ~27 flop per box (a wide node holds four), triangle test ~57
    alg_flops = {
        "extend": 108.0 * wc["nodesVisitedExtend"] + 57.0 * wc["trisTestedExtend"],
        "shadow": 108.0 * (wc["nodesVisited"] - wc["nodesVisitedExtend"]) + 57.0 * (wc["trisTested"] - wc["trisTestedExtend"]),
    }
    ktimes = {k: v for k, v in ktimes.items() if k in alg_bytes}
    # the dominant family is the one that takes longest when every kernel has the GPU to itself: durations measured while
    # batches overlap are stretched by whatever ran beside them, which makes two families of similar weight trade places
    rank = {k: v for k, v in (ktimes_alone or {}).items() if k in ktimes and v[0] > 0} or ktimes
    dominant = max(rank, key=lambda k: rank[k][0])
    k_ms, k_launches = ktimes[dominant]
    k_s = max(k_ms * 1e-3, 1e-12)
    launch_s = k_s / max(1e-9, k_launches)
    alg_gbps = alg_bytes[dominant] / k_s / 1e9
    counters, stamp = {}, {}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            doc = json.load(open(tpath)).get(workload, {})
            counters = doc.get(dominant, {}) or {}
            stamp = doc.get("_stamp", {}) or {}
        except Exception:
            counters = {}
    if not isinstance(counters, dict):
        counters = {}
    sha_now = kernel_sources_sha16()
    counters_stale = None
    if counters:
        counters_stale = (None if stamp.get("kernel_sources_sha16") == sha_now else
                          "counters were collected on other kernel sources (%s) than the ones benched (%s): traffic / valu figures are indicative only"
                          % (stamp.get("kernel_sources_sha16", "unstamped"), sha_now))
    traffic = counters.get("hbm_bytes_per_launch")
    valu_insts = counters.get("valu_insts_per_launch")
    hbm_frac = (traffic / launch_s / 1e9 / HBM_PEAK_GBS) if traffic else None
    valu_ginst = (valu_insts / launch_s / 1e9) if valu_insts else None
    valu_frac = (valu_ginst / VALU_PEAK_GINST) if valu_ginst is not None else None
    cache_note = None
    # no counter data for this workload: the traversal families are VALU-bound wherever they were profiled, so
    # that is the bound named (with no fraction: it needs the counters); the streaming kernels are HBM-bound
    bound = "valu" if dominant in alg_flops else "hbm"
    if hbm_frac is not None and valu_frac is not None:
        bound = "hbm" if hbm_frac >= valu_frac else "valu"
    if scene_bytes is not None and scene_bytes < 32 * 2**20 and dominant in alg_flops:
        # nodes + triangles fit the 32 MiB of L2: every record fetch is a cache hit; the kernel is instruction-bound
        bound = "valu"
        cache_note = "scene records (%.1f MiB) are L2-resident: the algorithmic byte figure counts cache hits and is not an HBM measure" % (scene_bytes / 2**20)
    flops = alg_flops.get(dominant)

    def fp32_block(seconds):
        if not flops:
            return None
        tf = flops / seconds / 1e12
        return {"algorithmic_flops_per_launch": round(flops / max(1e-9, k_launches), 1), "achieved_TFLOPs": round(tf, 3), "peak_TFLOPs": round(FP32_PEAK_TFLOPS, 1),
                "frac_of_fp32_peak": round(tf / FP32_PEAK_TFLOPS, 4),
                # wave instructions x 64 lanes per algorithmic flop; 0.5 for a kernel made of nothing but full-wave FMAs
                "lane_ops_per_algorithmic_flop": round(valu_insts * k_launches * 64.0 / flops, 3) if valu_insts else None,
                "definition": "27 flop x 4 boxes per wide node + 57 per triangle test (SURVEY 8d), nodes / triangles from the device's counters"}
    alone = None
    if ktimes_alone and ktimes_alone.get(dominant, (0, 0))[0] > 0:
        a_ms, a_launches = ktimes_alone[dominant]
        a_s = a_ms * 1e-3
        a_launch_s = a_s / max(1, a_launches)
        a_fp = fp32_block(a_s)
        alone = {"avg_launch_ms": round(a_ms / max(1, a_launches), 5),
                 "frac": (round(valu_insts / a_launch_s / 1e9 / VALU_PEAK_GINST, 4) if valu_insts else None) if bound == "valu"
                 else round(alg_bytes[dominant] / a_s / 1e9 / HBM_PEAK_GBS, 4),
                 "algorithmic_GBps": round(alg_bytes[dominant] / a_s / 1e9, 2),
                 "frac_of_hbm_peak_counters": round(traffic / a_launch_s / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                 "frac_of_valu_peak": round(valu_insts / a_launch_s / 1e9 / VALU_PEAK_GINST, 4) if valu_insts else None,
                 "frac_of_fp32_peak": a_fp["frac_of_fp32_peak"] if a_fp else None,
                 "kernel_ms_per_step": {k: round(v[0], 4) for k, v in ktimes_alone.items() if k in alg_bytes},
                 "note": "one batch at a time, one stream: every kernel has the GPU to itself"}
    if bound == "valu":
        achieved, peak, unit = valu_ginst, VALU_PEAK_GINST, "G wave64-VALU-instr/s"
        frac = valu_frac
    else:
        achieved, peak, unit = alg_gbps, HBM_PEAK_GBS, "GB/s"
        frac = alg_gbps / HBM_PEAK_GBS if not cache_note else None
    return {"bound": bound, "kernel": KERNEL_NAMES[dominant], "achieved": round(achieved, 2) if achieved is not None else None, "peak": peak, "unit": unit,
            "frac": round(frac, 4) if frac is not None else None, "traffic": traffic,
            "fp32": fp32_block(k_s), "alone": alone,
            "algorithmic_GBps": round(alg_gbps, 2), "algorithmic_bytes_per_launch": round(alg_bytes[dominant] / max(1e-9, k_launches), 1),
            "hbm_counters": {"bytes_per_launch": traffic, "GB/s": round(traffic / launch_s / 1e9, 1) if traffic else None,
                             "frac_of_hbm_peak": round(hbm_frac, 4) if hbm_frac is not None else None,
                             "l2_hit_rate": counters.get("l2_hit_rate")},
            "valu": {"wave_insts_per_launch": valu_insts, "Ginst/s": round(valu_ginst, 1) if valu_ginst else None,
                     "peak_Ginst/s": VALU_PEAK_GINST, "frac_of_valu_peak": round(valu_frac, 4) if valu_frac is not None else None,
                     "salu_insts_per_launch": counters.get("salu_insts_per_launch"),
                     "lane_utilisation": counters.get("valu_lane_utilisation"),
                     # SQ_ACTIVE_INST_VALU (quad-cycles a SIMD spends executing VALU instructions) x 4 / SIMD cycles, kernels alone
                     # under rocprofv3 --pmc: the share of time the vector ALU port is taken.  frac_of_valu_peak prices every
                     # instruction at the 2 clocks of an fp32 FMA; conversions, integer, select and 3-operand instructions --
                     # most of a node step -- take 4 (tools/ubench/valu_asm.hip), which this counter sees
                     "busy_under_pmc": counters.get("valu_busy_under_pmc")},
            "counters_commit": stamp.get("commit"), "counters_kernel_sha16": stamp.get("kernel_sources_sha16"), "benched_kernel_sha16": sha_now,
            "counters_warning": counters_stale,
            "avg_launch_ms": round(k_ms / max(1e-9, k_launches), 5), "launches_per_step": round(k_launches, 2),
            "nodes_per_extend_ray": round(wc["nodesVisitedExtend"] / max(1, n_ext), 3),
            "tris_per_extend_ray": round(wc["trisTestedExtend"] / max(1, n_ext), 3),
            "kernel_ms_per_step": {k: round(v[0], 4) for k, v in ktimes.items()},
            "scene_record_bytes": scene_bytes, "cache_note": cache_note,
            "note": "kernel durations are HIP-event pairs around every launch while batches overlap on two stream sets, as in the timed steps; "
                    "traffic / valu counters come from profiles/traffic.json (rocprofv3 --pmc passes of this command), stamped with the kernel sources they were taken on"}


def cpu_baseline(spec, wl, flat, seconds):
    """The oracle (restated src/cpu ThreadPool path) on this box's host cores, on a bounded sample of the same workload."""
    from tests import scenes as oracle_side  # the checker / baseline (imports oracle/), never the thing measured above
    workers, cores, cores_source = cpu_workers()
    spec.__class__ = oracle_side.SceneSpec
    oracle = spec.build_oracle()
    bw, bh, spp, max_depth = wl["width"], wl["height"], wl["spp"], wl["max_depth"]
    if wl["flat_cpu"]:
        oracle.set_flat(flat["model_nodes"], flat["nodes"], flat["tris"], flat["info_bytes"])
    ocam = spec.oracle_camera()
    # calibrate on a 1/16 window, then size the sample to the time budget
    win = (bw // 2 - bw // 8, bh // 2 - bh // 8, bw // 2 + bw // 8, bh // 2 + bh // 8)
    t1 = time.perf_counter()
    res = oracle.render(ocam, bw, bh, spp, max_depth=max_depth, window=win, samples=(0, 1), workers=workers, flat=wl["flat_cpu"])
    cal = time.perf_counter() - t1
    full_pass_est = cal * 16.0
    if full_pass_est <= seconds:
        passes = int(max(1, min(spp, seconds // max(full_pass_est, 1e-3))))
        reps = int(max(1, min(64, seconds // max(full_pass_est * passes, 1e-3))))
        t1 = time.perf_counter()
        for _ in range(reps):
            res = oracle.render(ocam, bw, bh, spp, max_depth=max_depth, samples=(0, passes), workers=workers, flat=wl["flat_cpu"])
        cpu_s = (time.perf_counter() - t1) / reps
        sample = f"{passes} of {spp} sample passes of the full {bw}x{bh} frame, timed {reps}x"
    else:
        reps = int(max(1, min(16, seconds // max(cal, 1e-3))))
        t1 = time.perf_counter()
        for _ in range(reps):
            res = oracle.render(ocam, bw, bh, spp, max_depth=max_depth, window=win, samples=(0, 1), workers=workers, flat=wl["flat_cpu"])
        cpu_s = (time.perf_counter() - t1) / reps
        sample = f"1 sample pass of the centre {win[2] - win[0]}x{win[3] - win[1]} window of the {bw}x{bh} frame, timed {reps}x"
    oc = res["counters"]
    cpu_rays = oc["extendPrimary"] + oc["extendSecondary"] + oc["shadow"]
    return {"value": round(cpu_rays / cpu_s / 1e6, 3), "unit": "Mrays/s", "cores": workers, "host_cores_available": cores, "host_cores_source": cores_source, "kind": "port",
            "sample": sample + (" (variant B: CPU walk of the same flattened BVH)" if wl["flat_cpu"] else
                                " (variant A: reference-faithful model tree + linear triangle scan)") +
            f", ThreadPool restatement with {workers} workers (hardware_concurrency()-2 of the {cores} cores this job may use), g++ -O2 -ffp-contract=off",
            "seconds": round(cpu_s, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="hf10m")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dynamic", action="store_true", help="N>1: tiles from a shared cursor (dynamic rebalancing) instead of static round-robin")
    ap.add_argument("--tile-batch", type=int, default=16, help="--dynamic: tiles taken from the cursor at a time")
    ap.add_argument("--no-roofline", action="store_true", help="skip the extra instrumented frames (used under rocprofv3 --pmc)")
    ap.add_argument("--family-times", action="store_true",
                    help="with --no-roofline: still run the HIP-event timing frames (frames like the timed ones) and print each kernel family's "
                         "average launch duration, so that a rocprofv3 --kernel-trace of THIS process can be set beside them")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget for the bounded CPU-baseline sample")
    ap.add_argument("--bvh", choices=("device", "host"), default="device",
                    help="which builder makes the triangle-level BVH of the timed frames: the device's (Morton LBVH + SAH leaves, "
                         "milliseconds) or the host's binned SAH (seconds); the other one is rendered untimed afterwards as the A/B")
    ap.add_argument("--torch-gather", action="store_true", help="N>1: gather with torch.distributed instead of rthip_gather (the C ABI's RCCL gather)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` by itself: this process becomes the launcher (it touches no GPU) and starts N fresh
        # ranks of this same script, one per GPU, exactly as `python -m torch.distributed.run --nproc-per-node N` would
        from raytracer_amd import dist as rdist
        try:
            rc, text = rdist.launch_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus,
                                          backend=os.environ.get("RT_DIST_BACKEND", "nccl"))
        except RuntimeError as e:
            raise SystemExit("bench.py: %s" % e)
        sys.stdout.write(text)
        sys.stdout.flush()
        raise SystemExit(rc)

    import torch
    import torch.distributed as dist

    import raytracer_amd as ra
    from raytracer_amd import dist as rdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # RT_DIST_BACKEND=gloo lets the N>1 code path be rehearsed with several ranks sharing one GPU (RCCL refuses two ranks
    # on one device); the driver's real runs use nccl (= RCCL over xGMI), one rank per GPU
    backend = os.environ.get("RT_DIST_BACKEND", "nccl")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP backend has no CPU fallback")
    local_rank = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    if world > 1:
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    n_gpus = max(args.gpus, world) if world > 1 else 1

    wl = workloads()[args.workload]
    spec = wl["spec"]()
    width, height = wl["width"], wl["height"]  # the same frame at every N: strong scaling
    spp, max_depth, ts = wl["spp"], wl["max_depth"], 64

    device_built = args.bvh == "device"
    r = spec.build_product(options=dict(maxDepth=max_depth, deviceBvh=1 if device_built else 0), device=local_rank)
    build_host_s, build_device_ms = r.build_times()
    flat = r.flat()
    num_tris, num_nodes = int(flat["info"].numTris), int(flat["info"].numNodes)
    assignment = rdist.assign_tiles(width, height, world, ts)
    if world > 1:
        r.set_tiles(assignment[rank])
    cam = spec.product_camera()
    hl = ra.capi.hip_lib()
    ctx = r.context
    dev = f"cuda:{local_rank}"
    n_local_floats = len(assignment[rank]) * ts * ts * 3 if world > 1 else None
    full_frame = torch.zeros((height, width, 3), dtype=torch.float32, device=dev) if (world > 1 and rank == 0) else None
    gather_ms = []
    tiles_total = int(sum(len(a) for a in assignment))
    dynamic = args.dynamic and world > 1
    # the C ABI's own RCCL gather (rthip_gather) unless rehearsing over gloo or asked otherwise
    native_gather = world > 1 and backend == "nccl" and not args.torch_gather
    per_rank_floats = rdist.padded_tile_count(assignment) * ts * ts * 3
    gathered = None
    gather_note = None
    if native_gather:
        try:
            rdist.init_native_comm(r, dist, rank, world)
        except ra.RtError as e:  # raised on every rank alike: RCCL could not be loaded or joined from the C ABI
            native_gather = False
            gather_note = "rthip_gather unavailable (%s): gathered with torch.distributed" % e
            if rank == 0:
                print("bench: " + gather_note, file=sys.stderr, flush=True)
    dyn_ids = dyn_counts = None
    if native_gather:
        if rank == 0 and not dynamic:
            gathered = torch.empty((world, per_rank_floats), dtype=torch.float32, device=dev)
        if rank == 0 and dynamic:  # rthip_gather_tiles: every rank's tiles, one after the other, whoever rendered them
            gathered = torch.empty((tiles_total * ts * ts * 3,), dtype=torch.float32, device=dev)
        dyn_ids, dyn_counts = np.zeros(tiles_total, np.uint32), np.zeros(world, np.uint32)

    queue = None
    local_buf = None
    if dynamic:
        queue = rdist.TileQueue(rdist.make_queue_store(dist, rank, world, os.environ.get("MASTER_ADDR", "127.0.0.1")), tiles_total, args.tile_batch)
        local_buf = torch.empty((tiles_total * ts * ts, 3), dtype=torch.float32, device=dev)  # a rank may end up with any number of tiles
    frame_no = [0]
    tile_counts, finish_times = [], []

    stage = recv = None
    pending = []
    if world > 1 and not dynamic and not native_gather:
        stage = [torch.zeros((per_rank_floats // 3, 3), dtype=torch.float32, device=dev) for _ in range(2)]
        if rank == 0 and backend == "nccl":
            recv = [[torch.empty_like(stage[0]) for _ in range(world)] for _ in range(2)]

    def untile_all(bufs):
        for buf, tiles in zip(bufs, assignment):
            rc = hl.rthip_untile(ctx, buf.data_ptr(), ra.capi._up(tiles), len(tiles), full_frame.data_ptr())
            assert rc == 0, ra.capi.error_string(rc)

    def drain():
        """Completes the torch gather in flight (if any); rank 0 scatters the tiles it brought into the row-major frame."""
        while pending:
            work, dst, _src = pending.pop(0)
            work.wait()
            if backend == "nccl":
                torch.cuda.current_stream().synchronize()
            if rank == 0:
                untile_all([b if backend == "nccl" else b.to(dev) for b in dst])

    def step():
        if not dynamic:
            # frames are queued back to back on the context's streams (no host round trip inside a frame or between
            # frames); the timed region ends with a synchronisation, and a multi-GPU step waits below when it gathers
            r.begin_frame(cam, width, height, spp)
            r.enqueue_passes(0, spp)
        if native_gather and not dynamic:
            # rthip_gather: the rank's tile-major accumulator goes to rank 0 over RCCL (grouped send/recv), queued on the
            # context's stream behind the frame; rank 0 then re-tiles the gathered buffers into the row-major frame
            t0 = time.perf_counter()
            rc = hl.rthip_gather(ctx, gathered.data_ptr() if rank == 0 else None, per_rank_floats, 0)
            assert rc == 0, ra.capi.error_string(rc)
            if rank == 0:
                untile_all([gathered[k] for k in range(world)])
            gather_ms.append((time.perf_counter() - t0) * 1e3)
        elif world > 1 and not dynamic:
            # torch.distributed path: the gather of frame k travels while frame k+1 renders (two staging buffers)
            t0 = time.perf_counter()
            k = frame_no[0] & 1
            frame_no[0] += 1
            rc = hl.rthip_copy_accumulator_to(ctx, stage[k].data_ptr(), n_local_floats)
            assert rc == 0, ra.capi.error_string(rc)
            drain()
            src = stage[k] if backend == "nccl" else stage[k].cpu()
            dst = None
            if rank == 0:
                dst = recv[k] if backend == "nccl" else [torch.empty_like(src) for _ in range(world)]
            pending.append((dist.gather(src, dst, dst=0, async_op=True), dst, src))
            gather_ms.append((time.perf_counter() - t0) * 1e3)
        if dynamic:
            # Tiles come from the shared cursor in batches and are queued on the context's two slots as they come
            # (rthip_enqueue_tiles: no host synchronisation); a rank draws again only when fewer than two of its batches are
            # unfinished (rthip_wait_tiles), which is what sends the work to whoever is fast.  The tiles' pixels line up in the
            # context's accumulator in arrival order; counters are read once per frame.
            queue.new_frame(frame_no[0])
            frame_no[0] += 1
            r.begin_dynamic(cam, width, height, spp, max_depth)
            while (batch := queue.next_batch()) is not None:
                r.enqueue_tiles(batch, max_in_flight=2)
            rc = hl.rthip_wait_tiles(ctx, 0)
            assert rc == 0, ra.capi.error_string(rc)
            finish_times.append(time.perf_counter())
            mine = r.local_tiles()
            tile_counts.append(len(mine))
            t0 = time.perf_counter()
            if native_gather:
                # rthip_gather_tiles: header [count | tile ids] + exactly the pixels each rank holds, grouped ncclSend / ncclRecv
                rc = hl.rthip_gather_tiles(ctx, gathered.data_ptr() if rank == 0 else None, gathered.numel() if rank == 0 else 0,
                                           ra.capi._up(dyn_ids), tiles_total, ra.capi._up(dyn_counts), 0)
                assert rc == 0, ra.capi.error_string(rc)
                if rank == 0:
                    n_all = int(dyn_counts.sum())
                    assert n_all == tiles_total, "every tile is rendered exactly once: %d of %d arrived" % (n_all, tiles_total)
                    rc = hl.rthip_untile(ctx, gathered.data_ptr(), ra.capi._up(dyn_ids), n_all, full_frame.data_ptr())
                    assert rc == 0, ra.capi.error_string(rc)
            else:
                # rehearsal over gloo (several ranks on one GPU): the same frame through torch.distributed
                rc = hl.rthip_copy_accumulator_to(ctx, local_buf.data_ptr(), tiles_total * ts * ts * 3)
                assert rc == 0, ra.capi.error_string(rc)
                local = local_buf[: len(mine) * ts * ts]
                got = rdist.gather_dynamic(local if backend == "nccl" else local.cpu(), mine, rank, world, dist, ts * ts)
                if rank == 0:
                    for tiles, src in got:
                        if len(tiles) == 0:
                            continue
                        src = src.to(dev).contiguous()
                        rc = hl.rthip_untile(ctx, src.data_ptr(), ra.capi._up(tiles), len(tiles), full_frame.data_ptr())
                        assert rc == 0, ra.capi.error_string(rc)
                torch.cuda.synchronize()
            gather_ms.append((time.perf_counter() - t0) * 1e3)

    def sync_all():
        drain()
        hl.rthip_synchronize(ctx)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1:
        # every rank would measure its traversal mode on its own tiles: rank 0's choice is used by all (same balance everywhere)
        r.begin_frame(cam, width, height, spp)
        rdist.agree_traversal_mode(r, dist, rank, world)
    for _ in range(args.warmup):
        step()
    gather_ms.clear()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    cnt = r.counters()  # totals of the LAST frame (begin_frame resets them)
    _mode, _cal = r.refill_mode()
    _grid = r.grid_blocks()
    rays_local = cnt["extendPrimary"] + cnt["extendSecondary"] + cnt["shadow"]
    t = torch.tensor([elapsed, float(rays_local), float(cnt["extendPrimary"]), float(cnt["extendSecondary"]), float(cnt["shadow"])],
                     dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
    rays_frame = float(t[1])
    ms_per_step = elapsed / args.steps * 1e3
    balance = None
    if dynamic:
        # per-rank tile counts of the last step and the idle tail: how long the first rank to run dry waited for the last
        mine_stats = [tile_counts[-1] if tile_counts else 0, finish_times[-args.steps:] if finish_times else []]
        everyone = [None] * world
        dist.all_gather_object(everyone, mine_stats)
        ends = np.array([e[1] for e in everyone], dtype=np.float64)  # [rank][step]
        balance = {"tiles_per_rank_last_step": [int(e[0]) for e in everyone], "tile_batch": args.tile_batch, "batches_in_flight_per_rank": 2,
                   "idle_tail_ms_mean": round(float(np.mean(ends.max(0) - ends.min(0)) * 1e3), 4) if ends.size else None}
        r.set_tiles(assignment[rank])
    value = rays_frame / (ms_per_step * 1e-3) / 1e6

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- the assembled multi-GPU frame must be the single-GPU frame, bit for bit (untimed: rank 0 renders all tiles itself)
    frame_check = None
    if world > 1:
        r.set_tiles(np.arange(tiles_total, dtype=np.uint32))
        r.begin_frame(cam, width, height, spp)
        r.render_passes(0, spp)
        same = np.array_equal(r.accumulator().view(np.uint32), full_frame.cpu().numpy().view(np.uint32))
        frame_check = "assembled frame bit-identical to a single-GPU render" if same else "MISMATCH against the single-GPU render"
        r.set_tiles(assignment[rank])

    if args.no_roofline:
        fam = None
        if args.family_times and world == 1:
            T = 3 if ms_per_step > 50 else 5
            r.set_option("refillMode", _mode)
            r.set_option("kernelTiming", 1)
            for _ in range(T):
                r.begin_frame(cam, width, height, spp)
                r.enqueue_passes(0, spp)
            fam = {k: {"avg_launch_ms": round(ms / max(1, n), 5), "launches_per_frame": round(n / T, 2)} for k, (ms, n) in r.kernel_times().items() if n}
            r.set_option("kernelTiming", 0)
        print(json.dumps({"metric": "Mrays/s (primary+secondary)", "value": round(value, 2), "unit": "Mrays/s", "ms_per_step": round(ms_per_step, 4),
                          "hip_event_family_times": fam}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel: per-kernel HIP-event timing + algorithmic work counts, in separate untimed frames.
    # Event pairs around every launch, resolved after the streams have drained: the frames below run exactly like the timed
    # ones (batches and frames overlapping), so a kernel's average duration here is what rocprofv3 reports for it.
    T = 3 if ms_per_step > 50 else 5
    r.set_option("refillMode", _mode)  # the instrumented frames run in the traversal mode the timed ones ran in (no new measurement)
    r.set_option("kernelTiming", 1)
    for _ in range(T):
        r.begin_frame(cam, width, height, spp)
        r.enqueue_passes(0, spp)
    ktimes = {k: (ms / T, launches / T) for k, (ms, launches) in r.kernel_times().items()}
    r.set_option("kernelTiming", 0)
    # the same kernels with the GPU to themselves (one slot, one stream): what each achieves when nothing shares the chip
    r.set_option("pipelining", 0)
    r.set_option("overlapShadow", 0)
    r.set_option("kernelTiming", 1)
    r.begin_frame(cam, width, height, spp)
    r.render_passes(0, spp)
    ktimes_alone = dict(r.kernel_times())
    r.set_option("kernelTiming", 0)
    r.set_option("pipelining", 1)
    r.set_option("overlapShadow", 1)
    r.set_option("countWork", 1)
    r.begin_frame(cam, width, height, spp)
    r.render_passes(0, spp)
    wc = r.counters()
    r.set_option("countWork", 0)
    r.set_option("refillMode", -1)
    roofline = roofline_block(args.workload, ktimes, wc, spp, ktimes_alone, 64 * (num_tris + num_nodes))

    # ---- the same scene through the DEVICE-built BVH (rthip_upload_scene_build): build time and frame time, untimed extra
    # ---- the same scene through the OTHER builder (untimed extra): its build time and frame time, the A/B of --bvh
    def builder_block(device, host_s, device_ms, nodes, ms):
        return {"builder": "device: Morton LBVH (bits dealt longest-extent-first) + SAH leaves, rthip_upload_scene_build" if device
                else "host: binned SAH, BvhFlattener.cpp", "build_ms_on_device": round(device_ms, 3) if device else None,
                "host_seconds": round(host_s, 3), "bvh_nodes": int(nodes), "ms_per_step": round(ms, 4)}
    bvh_used = builder_block(device_built, build_host_s, build_device_ms, num_nodes, ms_per_step)
    bvh_other = None
    if world == 1:
        flat = dict(flat, model_nodes=flat["model_nodes"].copy(), nodes=flat["nodes"].copy(), tris=flat["tris"].copy(), _owner=None)
        r.close()
        r2 = spec.build_product(options=dict(maxDepth=max_depth, deviceBvh=0 if device_built else 1), device=local_rank)
        host_s, device_ms = r2.build_times()
        f2 = r2.flat()["info"]
        for _ in range(2):
            r2.begin_frame(cam, width, height, spp)
            r2.enqueue_passes(0, spp)
        r2.synchronize()
        reps = max(3, args.steps)  # as many frames queued back to back as the timed region had: a shorter run carries more of the pipeline's ramp and drain per frame
        t1 = time.perf_counter()
        for _ in range(reps):
            r2.begin_frame(cam, width, height, spp)
            r2.enqueue_passes(0, spp)
        r2.synchronize()
        bvh_other = builder_block(not device_built, host_s, device_ms, f2.numNodes, (time.perf_counter() - t1) / reps * 1e3)
        bvh_other["frame_time_ratio_used_over_this"] = round(ms_per_step / bvh_other["ms_per_step"], 4)
        r2.close()

    cpu = None
    if not args.no_cpu_baseline:
        cpu = cpu_baseline(spec, wl, flat, args.cpu_seconds)

    refill_info = {"mode": ("low refill thresholds (incoherent rays)", "high refill thresholds (coherent rays)", "packet traversal of camera and shadow rays",
                            "packet traversal of camera rays")[_mode],
                   "list_launch_blocks_per_cu": _grid,
                   "calibration_ms_low_high_packet": [round(float(v), 3) for v in _cal[:3]],
                   "note": "how a wave treats its 64 rays, measured once per scene and frame shape before the warm-up step (rthip_set_refill_mode)"}
    out = {"metric": "Mrays/s (primary+secondary)", "value": round(value, 2), "unit": "Mrays/s", "n_gpus": n_gpus, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": wl["name"], "width": width, "height": height, "spp": spp, "max_depth": max_depth,
                      "tris": num_tris, "bvh_nodes": num_nodes, "bvh": bvh_used, "bvh_other_builder": bvh_other,
                      "rays_per_step": int(rays_frame), "primary": int(t[2]), "secondary_extend": int(t[3]), "shadow": int(t[4]),
                      "parallelism": (f"tiles64-{'dynamic' if dynamic else 'rr'}-x{world}-{backend}-{'rthip_gather' if native_gather else 'torch_gather'}"
                                      if world > 1 else "single"),
                      "balance": balance, "gather_ms": round(float(np.mean(gather_ms)), 4) if gather_ms else None, "gather_note": gather_note, "frame_check": frame_check,
                      "refill": refill_info},
           "roofline": roofline, "cpu_baseline": cpu}
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
