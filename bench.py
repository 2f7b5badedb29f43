#!/usr/bin/env python3
"""Benchmark of the ray-tracing hot path on MI355X.

One step = one frame of the workload: ray generation, kd-tree closest-hit
passes, shadow + shading passes and the colour fold, with the scene, the tree
and all work buffers already resident in HBM.  With --gpus N the same frame is
split into bands of 16 rows dealt round-robin to the N ranks (scene
replicated) and the finished rows are gathered on rank 0 with one RCCL
collective, inside the timed region ("strong" scaling: the frame is fixed).

Prints ONE JSON line on rank 0.  `value` is whole-job Mrays/s, a ray being
one closest-hit query or one shadow query (counted by the kernels' own work
lists, not estimated).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# A frame runs on three streams of its own (closest-hit chain + two auxiliary ones); with the
# collective library's streams on top a rank would exceed the runtime's default of four
# hardware queues per process and two of them would be folded into one queue, which
# serialises what is meant to overlap.  Must be set before the HIP runtime starts.
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch
import torch.distributed as dist

from raytracing_amd import dist as rdist
from raytracing_amd import scenes
from raytracing_amd.pipe import RT_HIP_STATS, RT_HIP_TIMING, RT_PARAM_OVERLAP, DeviceTree, FlatTree, HipRenderer

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, spec

# kd-tree of the benchmark: the reference's split rule with its two limits made
# run-time parameters (the reference's fixed depth 10 leaves ~2000 triangles
# per leaf on this scene)
TREE_DEPTH, TREE_LEAF = 26, 12

WORKLOADS = {
    "config3": dict(make=scenes.config3, text="1M-triangle displaced sphere + floor, 3840x2160, "
                    "opaque reflective material, 5 rays per pixel (primary + 4 bounces), 1 point light"),
    "config2": dict(make=scenes.config2, text="100k-triangle displaced sphere + floor, 1920x1080, "
                    "primary + shadow rays, 1 point light"),
}


def algorithmic_bytes(nodes, tris, primary, hits):
    """8 B per kd node visited + 48 B per triangle tested + 4 B frame-buffer
    write per primary ray + 128 B of shading data per shaded hit."""
    return 8 * nodes + 48 * tris + 4 * primary + 128 * hits


def measured_traffic(args, kernel, world):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 counter
    passes of this same command (profiles/r01_traffic.json: FETCH_SIZE doubled
    as the gfx950 correction prescribes, plus WRITE_SIZE), or None when that
    file does not cover this configuration.  Counters cannot be read live."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
    except (OSError, ValueError):
        return None
    key = f"{args.workload}/depth{args.depth}/leaf{args.leaf}/gpus{world}/{'serial' if args.no_overlap else 'overlap'}"
    return t.get(key, {}).get("k_" + kernel)


def cpu_baseline(sc, tree, seconds=15.0):
    """The CPU restatement (oracle, checker-only code) timed on a bounded
    lattice of the same frame and the same tree."""
    from oracle import orc
    nodes, idx = tree.expand()
    threads = min(os.cpu_count() or 1, 16)
    step, rays, dt = 64, 0, 1e-3
    for _ in range(4):
        t0 = time.time()
        _, _, st = orc.raytrace(sc, tree.bbox, nodes, idx, x_step=step, y_step=step, threads=threads, want_hits=False)
        dt = max(time.time() - t0, 1e-3)
        rays = st["raysClosest"] + st["raysShadow"]
        if dt >= 0.5 * seconds or step == 1:
            break
        # denser lattice so that the run takes about `seconds`
        step = int(max(1, min(step - 1, step * (dt / seconds) ** 0.5)))
    return {"value": round(rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": ("the whole frame" if step == 1 else f"every {step}th pixel of every {step}th row of the same frame")
                      + f" on the same tree ({rays} rays in {dt:.1f} s on {threads} threads)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="config3", choices=sorted(WORKLOADS))
    ap.add_argument("--depth", type=int, default=TREE_DEPTH)
    ap.add_argument("--leaf", type=int, default=TREE_LEAF)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run every pass of a frame on one stream, one after the other (kernel-isolated timings)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    if os.environ.get("RT_BENCH_WATCHDOG"):          # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["RT_BENCH_WATCHDOG"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # one rank per GPU; on a box with fewer GPUs than ranks (rehearsals) ranks share devices
    local_dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl") needs one GPU per rank; RT_DIST_BACKEND=gloo rehearses the multi-rank path on one GPU
        backend = os.environ.get("RT_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    wl = WORKLOADS[args.workload]
    sc = wl["make"]()
    v, t = sc.flatten()
    t0 = time.time()
    tree = FlatTree.build(v, t, args.depth, args.leaf)
    build_s = time.time() - t0

    # the same tree built on the GPU (outside the timed region; reported, and checked against the host tree)
    DeviceTree.build(v, t, args.depth, args.leaf).release()
    t0 = time.time()
    dtree = DeviceTree.build(v, t, args.depth, args.leaf)
    gpu_build_s = time.time() - t0
    assert (dtree.n_nodes, dtree.n_recs) == (len(tree.nodes), len(tree.recs))
    dtree.release()

    stream = torch.cuda.current_stream()
    r = HipRenderer(local_dev, stream=stream.cuda_stream)
    r.load_scene(sc, tree)
    r.set_param(RT_PARAM_OVERLAP, 0 if args.no_overlap else 1)
    r.set_tile(rdist.BAND_H, world, rank)
    rows = r.local_rows
    pad = rdist.padded_rows(sc.h, rdist.BAND_H, world)
    frame_local = torch.zeros((pad, sc.w, 4), dtype=torch.uint8, device=dev)

    dbg = bool(os.environ.get("RT_BENCH_TRACE"))     # debugging aid: synchronise and report after every frame

    def step(flags=0):
        r.render(flags, d_out=frame_local.data_ptr())
        if dbg:
            r.sync()
            print(f"[rank {rank}] frame rendered", file=sys.stderr, flush=True)
        if world > 1:
            out = rdist.gather_frame(frame_local, sc.h, sc.w, rdist.BAND_H)
            if dbg:
                print(f"[rank {rank}] frame gathered", file=sys.stderr, flush=True)
            return out
        return frame_local

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()

    # ---- timed region: exactly K steps ------------------------------------
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(RT_HIP_TIMING)
    fence()
    elapsed = time.perf_counter() - t0
    # per-kernel HIP-event times, averaged over the timed steps (the events are
    # recorded on the stream the kernels run on); read after the timed region
    kms, launches = r.kernel_times()
    st = r.stats()                      # ray counters of the last frame (identical for every frame)

    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    cnt = torch.tensor([st["raysClosest"], st["raysShadow"], st["hits"]], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    rays_frame = float(cnt[0] + cnt[1])

    # ---- outside the timed region: the same frame with every pass alone on the
    # chip (what a kernel takes when it does not share the GPU with the passes
    # that the normal schedule runs beside it)
    r.set_param(RT_PARAM_OVERLAP, 0)
    for _ in range(3):
        r.render(RT_HIP_TIMING, d_out=frame_local.data_ptr())
    r.sync()
    kms_iso, _ = r.kernel_times()
    r.set_param(RT_PARAM_OVERLAP, 0 if args.no_overlap else 1)

    # ---- roofline of the dominant kernel (counting pass, outside the timed region)
    r.render(RT_HIP_STATS, d_out=frame_local.data_ptr())
    r.sync()
    sst = r.stats()
    nodes_c = sst["nodeVisits"] - sst["nodeVisitsShadow"]
    tris_c = sst["triTests"] - sst["triTestsShadow"]
    primary = sc.w * rows
    kbytes = {
        # closest-hit passes: nodes + triangle records of the closest-hit queries
        "closest": 8 * nodes_c + 48 * tris_c,
        # shadow passes: nodes + triangle records of the shadow queries
        "shadow": 8 * sst["nodeVisitsShadow"] + 48 * sst["triTestsShadow"],
        # shading passes: 128 B of shading data per shaded hit
        "shade": 128 * sst["hits"],
        "fold": 4 * primary,
    }
    dominant = max(("closest", "shadow", "shade"), key=lambda k: kms[k])
    dom_bytes, dom_ms = kbytes[dominant], kms[dominant]
    n_launch = max(launches.get(dominant, 1), 1)
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    frame_bytes = algorithmic_bytes(sst["nodeVisits"], sst["triTests"], primary, sst["hits"])

    out = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = rays_frame * args.steps / elapsed / 1e6
        out = {
            "metric": "Mrays/s (primary+shadow) on 1M-tri scene at 4K" if args.workload == "config3"
                      else "Mrays/s (primary+shadow)",
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {wl['text']}",
                       "kd_tree": {"max_depth": args.depth, "max_prims_in_leaf": args.leaf, "depth": int(tree.depth),
                                   "nodes": int(len(tree.nodes)), "triangle_refs": int(len(tree.recs)),
                                   "host_build_s": round(build_s, 3), "gpu_build_s": round(gpu_build_s, 4)},
                       "schedule": "passes in order on one stream" if args.no_overlap else
                                   "closest-hit chain on the main stream, shadow + lighting passes on two auxiliary "
                                   "streams (they overlap the chain; per-kernel times include the time a pass shares the chip)",
                       "rays_per_frame": int(rays_frame), "closest_rays": int(cnt[0]), "shadow_rays": int(cnt[1]),
                       "parallelism": "single GPU" if world == 1 else
                                      f"frame split in bands of {rdist.BAND_H} rows over {world} GPUs, scene replicated, "
                                      "rows gathered on rank 0 (RCCL)"},
            "kernel_ms_per_frame_rank0": {k: round(x, 4) for k, x in kms.items()},
            "kernel_ms_per_frame_isolated_rank0": {k: round(x, 4) for k, x in kms_iso.items()},
            "roofline": {"bound": "hbm", "kernel": "k_" + dominant,
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": measured_traffic(args, dominant, world),
                         "launches_per_frame": n_launch,
                         "algorithmic_bytes_per_launch": int(dom_bytes / n_launch),
                         "avg_launch_ms": round(dom_ms / n_launch, 4),
                         "frame_algorithmic_bytes": int(frame_bytes),
                         "frame_achieved_GBs": round(frame_bytes / (kms["frame"] * 1e-3) / 1e9, 2),
                         "per_ray": {"nodes": round(sst["nodeVisits"] / max(sst["raysClosest"] + sst["raysShadow"], 1), 2),
                                     "tri_tests": round(sst["triTests"] / max(sst["raysClosest"] + sst["raysShadow"], 1), 2)}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc, tree, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    r.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
