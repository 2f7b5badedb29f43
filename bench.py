#!/usr/bin/env python3
"""Benchmark of the ray-tracing hot path on MI355X.

One step = one frame of the workload: ray generation, kd-tree closest-hit
passes, shadow + shading passes and the colour fold, with the scene, the tree
and all work buffers already resident in HBM.

  config3 (default) / config2   one fixed frame.  With --gpus N the frame is
      split into bands of 16 rows dealt round-robin to the N ranks (scene
      replicated) and the finished rows are gathered on rank 0 through the
      library's C-ABI (RCCL send / receive + de-interleave kernel), inside
      the timed region: "strong" scaling, the frame is fixed.
  config5                       the fly-through: step i of rank r renders
      camera (i * N + r) mod 240 of the orbit, whole frames, no exchange:
      "weak" scaling, every rank renders K frames.

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

# A frame runs on three streams (closest-hit chain + two auxiliary ones) and four to six frames are in
# flight; with the null stream (the kd-tree builder's) and the collective library's streams a rank
# uses up to twenty-one.  The runtime's default of four hardware queues per process folds several
# streams into one queue, which serialises what is meant to overlap: measured, a frame of two in
# flight costs 7.2 ms at 8 queues (one stream too few) and 6.8 ms from 10 on; six contexts on 16 queues
# render a 1/8 share in 0.83 ms, on 20 to 24 queues in 0.72 to 0.76 ms (a full frame: no difference).  More
# streams than queues and streams share queues; and once a process has brought 24 or more hardware queues
# into being, everything it runs afterwards is slower (eight contexts: 1/8 share 1.2 ms; the fly-through frame
# after the other workloads 40.7 against 27.3 ms) - hence 22: five contexts of three streams, the checker's
# three, the null stream and the collective library's.  The streams themselves are reused from workload to
# workload (stream_for below, and the library's own pool): the runtime deals queues to streams in the order
# of their creation, and streams that come and go end up sharing a queue with a neighbour.  Must be set
# before the HIP runtime starts (the machine exports 4, hence an assignment).
os.environ["GPU_MAX_HW_QUEUES"] = "22"

import numpy as np
import torch
import torch.distributed as dist

from raytracing_amd import dist as rdist
from raytracing_amd import scenes
from raytracing_amd.pipe import (RT_HIP_STATS, RT_HIP_TIMING, RT_PARAM_LAUNCH_WAVES, RT_PARAM_OVERLAP, DeviceTree, FlatTree,
                                 HipRenderer)

# roofs: HBM3E spec; L2 and Infinity Cache as measured chip-wide for gathered rows (the access
# shape of this path), the upper end of the ranges
HBM_PEAK_GBS = 8000.0
L2_PEAK_GBS = 18800.0
MALL_PEAK_GBS = 8600.0

# kd-tree of the benchmark: the reference's split rule with its two limits made
# run-time parameters (the reference's fixed depth 10 leaves ~2000 triangles
# per leaf on this scene)
TREE_DEPTH, TREE_LEAF = 26, 12
# Leaves keep only the triangles that reach their cell (nodes and split planes stay the reference's; the
# reference files a triangle under every leaf its bounding box overlaps).  Same frame, pixel for pixel and hit
# id for hit id (tests/test_parity_gpu.py: test_pruned_leaves_give_the_frame_of_the_reference_leaves), with a
# quarter fewer triangle tests per ray.  --prune-leaves 0 renders the reference's leaf lists.
PRUNE_LEAVES = 1e-5
ORBIT_FRAMES = 240
IN_FLIGHT = 4                         # frames in flight (contexts that share the resident scene)
# ... of a rank that renders a quarter of the frame or less (its passes are short lists that leave most of the chip
# idle: 1/8 share 0.85 / 0.76 / 0.76 / 0.74 ms with 4 / 5 / 6 / 7 in flight, 1/4 share 1.40 / 1.33 / 1.33 / 1.30 ms; half and
# full frames gain nothing beyond four)
IN_FLIGHT_SMALL = 5
SMALL_FRAME = 2_200_000               # pixels: a quarter of a 4K frame, a 1080p frame


def in_flight_for(pixels, parts):
    """Frames in flight for a rank that renders `pixels` pixels per frame as one of `parts` ranks: four; five for small
    frames; six for small frames on a single rank (no collective library whose streams want queues too: the 1080p
    workload 0.66 / 0.63 / 0.60 ms per step with 4 / 5 / 6)."""
    if pixels > SMALL_FRAME:
        return IN_FLIGHT
    return IN_FLIGHT_SMALL + (1 if parts == 1 else 0)

PROFILE_TAG = "profiles/r03"          # the committed profiler summaries the line refers to

_STREAMS = []


def stream_for(k):
    """The k-th frame in flight's stream: one set for the whole process, whatever workload is being rendered."""
    while len(_STREAMS) <= k:
        _STREAMS.append(torch.cuda.Stream())
    return _STREAMS[k]


WORKLOADS = {
    "config3": dict(make=scenes.config3, text="1M-triangle displaced sphere + floor, 3840x2160, "
                    "opaque reflective material, 5 rays per pixel (primary + 4 bounces), 1 point light"),
    "config2": dict(make=scenes.config2, text="100k-triangle displaced sphere + floor, 1920x1080, "
                    "primary + shadow rays, 1 point light"),
    "config5": dict(make=scenes.config5, text="fly-through: 1M-triangle displaced sphere + floor, 3840x2160, reflective + "
                    "refractive material (incoherent secondary rays), 6 rays per pixel, 1 point light, "
                    f"camera on a {ORBIT_FRAMES}-frame orbit, frame f on rank f mod N"),
}


def algorithmic_bytes(nodes, tris, primary, hits):
    """8 B per kd node visited + 48 B per triangle tested + 4 B frame-buffer
    write per primary ray + 128 B of shading data per shaded hit."""
    return 8 * nodes + 48 * tris + 4 * primary + 128 * hits


def union_length(spans):
    """Total length covered by a list of (start, end) intervals."""
    total, end = 0.0, float("-inf")
    for a, b in sorted(spans):
        lo = max(a, end)
        if b > lo:
            total += b - lo
            end = b
    return total


def committed_counters(args, kernel, world):
    """What the committed rocprofv3 counter passes of this same frame say about
    `kernel` (profiles/r03_counters.json, written by tools/refresh_profiles.py:
    fabric traffic per launch with the gfx950 FETCH_SIZE correction, and the
    ratios that show what binds the kernel), or None when that file does not
    cover this configuration.  Counters cannot be read live."""
    try:
        t = json.load(open(os.path.join(ROOT, PROFILE_TAG + "_counters.json")))
    except (OSError, ValueError):
        return None
    key = (f"{args.workload}/depth{args.depth}/leaf{args.leaf}{'/pruned' if args.prune_leaves > 0 else ''}/gpus{world}/"
           f"{'serial' if args.no_overlap else 'overlap'}")
    return t.get(key, {}).get("k_" + kernel)


def cpu_baseline(sc, v, t, tree, seconds):
    """The CPU restatement (oracle, checker-only code) timed on bounded lattices
    of the same frame: on the benchmark's tree, and on the tree the reference's
    own limits give (depth 10, leaves of at most 7 - rt_types.h:16-17), which on
    this scene leaves ~2000 triangles per leaf."""
    from oracle import orc
    threads = min(os.cpu_count() or 1, 16)

    def timed(nodes, idx, bbox, budget, step0):
        step, rays, dt = step0, 0, 1e-3
        for _ in range(4):
            t0 = time.time()
            _, _, st = orc.raytrace(sc, bbox, nodes, idx, x_step=step, y_step=step, threads=threads, want_hits=False)
            dt = max(time.time() - t0, 1e-3)
            rays = st["raysClosest"] + st["raysShadow"]
            if dt >= 0.5 * budget or step == 1:
                break
            step = int(max(1, min(step - 1, step * (dt / budget) ** 0.5)))     # denser lattice, about `budget` seconds
        nth = {2: "second", 3: "third"}.get(step, f"{step}th")
        what = "the whole frame" if step == 1 else f"every {nth} pixel of every {nth} row of the same frame"
        return rays / dt / 1e6, f"{what} ({rays} rays in {dt:.1f} s on {threads} threads)"

    nodes, idx = tree.expand()
    deep, deep_txt = timed(nodes, idx, tree.bbox, 0.6 * seconds, 64)
    ref_tree = FlatTree.build(v, t, 10, 7)
    rn, ri = ref_tree.expand()
    shallow, shallow_txt = timed(rn, ri, ref_tree.bbox, 0.4 * seconds, 256)
    ref_tree.release()
    return {"value": round(deep, 4), "unit": "Mrays/s", "cores": threads, "host_cpus": os.cpu_count(), "kind": "port",
            "sample": deep_txt + " on the benchmark's tree",
            "reference_limits": {"value": round(shallow, 5), "unit": "Mrays/s", "cores": threads,
                                 "tree": "depth 10, at most 7 triangles per leaf unless the depth limit is hit (the reference's "
                                         "MAX_DEPTH / MAX_PRIMS_IN_NODE): what the reference's algorithm costs on this scene",
                                 "sample": shallow_txt}}


def side_workload(name, local_dev, dev, depth, leaf, prune, steps, tree=None, plain=None):
    """A short run of another workload for the default line's `other_configs`: frames in flight as in the
    timed region of the main workload, as many warm-up steps, `steps` timed steps, every slot's last frame compared with the
    same camera's frame rendered alone from the reference's leaf lists.  tree / plain: trees to reuse (same geometry)."""
    wl = WORKLOADS[name]
    sequence = name == "config5"
    sc = wl["make"]()
    v, t = sc.flatten()
    own = tree is None
    if own:
        tree = FlatTree.build(v, t, depth, leaf, prune_leaves=prune)
        plain = FlatTree.build(v, t, depth, leaf) if prune > 0 else None
    n_fl = in_flight_for(sc.w * sc.h, 1)
    slots = []
    for k in range(n_fl):
        st = stream_for(k)
        r = HipRenderer(local_dev, stream=st.cuda_stream)
        if k == 0:
            r.load_scene(sc, tree)
        else:
            r.share_scene(slots[0][0], sc)
        r.set_param(RT_PARAM_LAUNCH_WAVES, 8)
        slots.append((r, torch.zeros((sc.h, sc.w, 4), dtype=torch.uint8, device=dev)))
    cam = lambda i: scenes.make_camera(sc.w, sc.h, world=scenes.orbit_world(i % ORBIT_FRAMES, ORBIT_FRAMES))
    last = {}

    def step(i):
        r, out = slots[i % n_fl]
        if sequence:
            r.set_camera(cam(i))
        r.render(0, d_out=out.data_ptr())
        last[i % n_fl] = i

    torch.cuda.synchronize()
    for i in range(n_fl):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n_fl, n_fl + steps):
        step(i)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # (the checker's streams only exist once the timed steps are over)
    checker = HipRenderer(local_dev)
    if plain is not None and len(plain.recs) > len(tree.recs):
        checker.load_scene(sc, plain)
    else:
        checker.share_scene(slots[0][0], sc)
    checker.set_param(RT_PARAM_OVERLAP, 0)
    ok = True
    want = torch.zeros_like(slots[0][1])
    for k, i in sorted(last.items()):
        if sequence:
            checker.set_camera(cam(i))
        checker.render(0, d_out=want.data_ptr())
        checker.sync()
        ok = ok and bool(torch.equal(slots[k][1], want))
    rays = 0
    r0 = slots[0][0]
    for i in range(n_fl, n_fl + steps) if sequence else [0]:
        if sequence:
            r0.set_camera(cam(i))
        r0.render(0, d_out=slots[0][1].data_ptr())
        st = r0.stats()
        rays += (st["raysClosest"] + st["raysShadow"]) * (1 if sequence else steps)
    out = {"workload": f"{name}: {wl['text']}", "steps": steps, "warmup": n_fl, "frames_in_flight": n_fl,
           "ms_per_step": round(elapsed / steps * 1e3, 4), "value": round(rays / elapsed / 1e6, 2), "unit": "Mrays/s",
           "rays_per_step": int(rays / steps), "frames_verified": ok,
           "kd_tree": {"max_depth": depth, "max_prims_in_leaf": leaf, "triangle_refs": int(len(tree.recs)),
                       "leaves": "pruned (checked against the reference's lists)" if prune > 0 else "the reference's lists"}}
    checker.close()
    for r, _ in reversed(slots):
        r.close()
    if own:
        tree.release()
        if plain is not None:
            plain.release()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="config3", choices=sorted(WORKLOADS))
    ap.add_argument("--depth", type=int, default=TREE_DEPTH)
    ap.add_argument("--leaf", type=int, default=TREE_LEAF)
    ap.add_argument("--prune-leaves", type=float, default=PRUNE_LEAVES,
                    help="margin (x scene size) of the leaf pruning of the host builder; 0 = the reference's leaf lists")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-proxy", action="store_true", help="skip the strong-scaling proxy (1-GPU runs of a frame workload)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short runs of the other two workloads that the default single-GPU line carries")
    ap.add_argument("--timed-only", action="store_true",
                    help="for profiler runs: every launch of the frame kernels belongs to the warm-up or the timed steps (no "
                         "frame-alone latency, no in-order pass, no proxy, no check of the pruned leaves), so a profiler's per-kernel average is the timed "
                         "region's avg_launch_ms")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run every pass of a frame on one stream, one after the other (kernel-isolated timings)")
    ap.add_argument("--launch-waves", type=int, default=0,
                    help="waves per CU one traversal pass occupies (1..16); 0 = 16 for one frame at a time, with frames in "
                         "flight 8, and 6 for a rank's share of a quarter frame or less")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--in-flight", type=int, default=0,
                    help="frames in flight: consecutive frames alternate between this many contexts (own streams and work "
                         "buffers, one shared scene), so a frame's sparse last passes overlap the next frame's first ones.  "
                         "0 = 4, and 5 for a rank that renders a quarter of the frame or less (22 hardware queues: a context "
                         "takes three)")
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
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl") needs one GPU per rank; RT_DIST_BACKEND=gloo rehearses the multi-rank path on one GPU
        backend = os.environ.get("RT_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    sequence = args.workload == "config5"
    wl = WORKLOADS[args.workload]
    sc = wl["make"]()
    v, t = sc.flatten()
    t0 = time.time()
    tree = FlatTree.build(v, t, args.depth, args.leaf, prune_leaves=args.prune_leaves)
    build_s = time.time() - t0
    ref_refs = len(tree.recs)
    if args.prune_leaves > 0:
        plain = FlatTree.build(v, t, args.depth, args.leaf)
        assert len(plain.nodes) == len(tree.nodes) and len(plain.recs) >= len(tree.recs)
        ref_refs = len(plain.recs)
        plain.release()

    # the same tree built on the GPU (outside the timed region; reported, and checked against the host tree)
    DeviceTree.build(v, t, args.depth, args.leaf).release()
    t0 = time.time()
    dtree = DeviceTree.build(v, t, args.depth, args.leaf)
    gpu_build_s = time.time() - t0
    assert (dtree.n_nodes, dtree.n_recs) == (len(tree.nodes), ref_refs)        # (the GPU builder keeps the reference's leaf lists)
    dtree.release()

    split = world > 1 and not sequence
    pad = rdist.padded_rows(sc.h, rdist.BAND_H, world) if split else sc.h
    def slots_for(parts):
        if args.in_flight:
            return max(1, min(args.in_flight, 7))
        return in_flight_for(sc.w * sc.h // (parts if not sequence else 1), parts)

    def waves_for(parts, contexts):
        """Waves per CU one traversal pass occupies: all 16 for a frame on its own; with frames in flight 8, and 6 for
        shares of a quarter frame and less (three in flight: full frame 5.53 / 5.40 / 5.43 ms with 6 / 8 / 10, 1/4 frame
        1.66 / 1.68 / 1.68 ms, 1/8 frame 1.04 / 1.12 / 1.09 ms)."""
        if args.launch_waves:
            return args.launch_waves
        return 16 if contexts <= 1 else (8 if parts <= 2 else 6)

    n_slots = slots_for(world if split else 1)

    # ---- one context per frame in flight: own stream, own work buffers, own output and exchange; the scene is
    # resident once and shared
    class Slot:
        pass

    slots = []

    def add_slot(where):
        k = len(where)
        sl = Slot()
        sl.stream = stream_for(k)                # (not the default stream: work on it would wait for, and hold up, the other frame)
        sl.r = HipRenderer(local_dev, stream=sl.stream.cuda_stream)
        if k == 0:
            sl.r.load_scene(sc, tree)
        else:
            sl.r.share_scene(where[0].r, sc)
        sl.r.set_param(RT_PARAM_OVERLAP, 0 if args.no_overlap else 1)
        # frames in flight: a pass takes half the wave slots of a CU or less, so that the passes of the other frames find
        # room beside it (5.38 against 5.76 ms per step; a frame alone would be faster with all 16: 6.7 against 7.4 ms)
        sl.r.set_param(RT_PARAM_LAUNCH_WAVES, waves_for(world if split else 1, n_slots))
        if split:
            sl.r.set_tile(rdist.BAND_H, world, rank)
        sl.frame_local = torch.zeros((pad, sc.w, 4), dtype=torch.uint8, device=dev)
        sl.comm, sl.frame_full = None, None
        where.append(sl)

    for k in range(n_slots):
        add_slot(slots)
    torch.cuda.synchronize()                     # buffers were zeroed on torch's default stream
    r = slots[0].r
    rows = r.local_rows
    frame_local = slots[0].frame_local

    # ---- the checker: a context of its own that renders one frame at a time, every pass in order on one stream with
    # nothing beside it, from the REFERENCE's leaf lists when the timed frames use pruned ones.  Before the timed region
    # it decides whether the pruned lists may be used at all (this rank's share of the first frame, every pixel compared
    # on the device); after it, every slot's last timed frame is compared with the checker's frame of the same camera.
    # The checker only exists while it is used: a context's streams take hardware queues, and with more streams in
    # the process than queues streams share queues and the frames in flight queue up behind each other (step time
    # 5.54 against 5.20 ms with the checker's three streams left open beside four contexts).
    plain = None
    if not args.timed_only and args.prune_leaves > 0 and len(tree.recs) < ref_refs:
        plain = FlatTree.build(v, t, args.depth, args.leaf)

    def open_checker():
        ck = HipRenderer(local_dev)
        if plain is not None:
            ck.load_scene(sc, plain)
        else:
            ck.share_scene(r, sc)
        ck.set_param(RT_PARAM_OVERLAP, 0)
        if split:
            ck.set_tile(rdist.BAND_H, world, rank)
        return ck

    def checker_frame(ck, cam=None):
        if cam is not None:
            ck.set_camera(cam)
        out = torch.zeros_like(frame_local)
        torch.cuda.synchronize()
        ck.render(0, d_out=out.data_ptr())
        ck.sync()
        return out

    if plain is not None:
        cam0 = scenes.make_camera(sc.w, sc.h, world=scenes.orbit_world(rank % ORBIT_FRAMES, ORBIT_FRAMES)) if sequence else None
        ck = open_checker()
        want = checker_frame(ck, cam0)
        ck.close()
        if cam0 is not None:
            r.set_camera(cam0)
        got = torch.zeros_like(frame_local)
        torch.cuda.synchronize()
        r.render(0, d_out=got.data_ptr())
        r.sync()
        if not torch.equal(got, want):
            raise SystemExit("the pruned leaf lists change this frame: run with --prune-leaves 0")
        del got, want

    # ---- frame exchange of the split frame: the library's own RCCL gather (one communicator per frame in flight);
    # torch.distributed's gather (also RCCL) only where that cannot run (gloo rehearsals on one GPU) or fails to
    # start, and the JSON line says which
    gather_kind = None
    if split:
        if backend == "nccl" and os.environ.get("RT_BENCH_GATHER", "abi") != "torch":
            # Opening a communicator is a collective: a rank that cannot take part (no RCCL library to open) must be
            # known BEFORE the others enter it, or they would wait for it for ever.  So every rank first asks the
            # library for an id (that opens RCCL and calls into it, nothing collective), and all ranks compare notes.
            why = ""
            try:
                rdist.HipComm.unique_id()
            except Exception as e:                                     # noqa: BLE001 - reported in the JSON line
                why = str(e)
            ok = torch.tensor([0 if why else 1], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()):
                for sl in slots:
                    idt = torch.zeros(rdist.HipComm.ID_BYTES, dtype=torch.uint8, device=dev)
                    if rank == 0:
                        idt.copy_(torch.frombuffer(bytearray(rdist.HipComm.unique_id()), dtype=torch.uint8))
                    dist.broadcast(idt, 0)
                    try:
                        sl.comm = rdist.HipComm(local_dev, world, rank, bytes(idt.cpu().numpy().tobytes()))
                    except Exception as e:                             # noqa: BLE001
                        why = str(e)
                    sl.frame_full = torch.zeros((sc.h, sc.w, 4), dtype=torch.uint8, device=dev) if rank == 0 else None
                ok = torch.tensor([1 if all(sl.comm for sl in slots) else 0], device=dev)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()):
                gather_kind = "rt_hip_gather_frame (C-ABI: RCCL send/receive to rank 0 + de-interleave kernel)"
            else:
                for sl in slots:
                    if sl.comm:
                        sl.comm.close()
                    sl.comm = None
                gather_kind = ("torch.distributed.gather over RCCL (the C-ABI communicator could not be opened on every rank"
                               + (f"; this rank: {why}" if why else "") + ")")
        elif backend == "nccl":
            gather_kind = "torch.distributed.gather over RCCL (RT_BENCH_GATHER=torch)"
        else:
            gather_kind = f"torch.distributed.gather over {backend} (rehearsal: ranks share a GPU, which RCCL refuses)"
    use_comm = split and all(sl.comm for sl in slots)
    all_slots, slots = slots, slots[:n_slots]

    trace = bool(os.environ.get("RT_BENCH_TRACE"))     # debugging aid: synchronise and report after every frame

    def camera_of(step_index):
        return scenes.make_camera(sc.w, sc.h, world=scenes.orbit_world((step_index * world + rank) % ORBIT_FRAMES, ORBIT_FRAMES))

    def step(i, flags=0):
        sl = slots[i % n_slots]
        sl.last_step = i
        if sequence:
            sl.r.set_camera(camera_of(i))
        sl.r.render(flags, d_out=sl.frame_local.data_ptr())
        if trace:
            sl.r.sync()
            print(f"[rank {rank}] frame rendered", file=sys.stderr, flush=True)
        if not split:
            sl.last_out = sl.frame_local
        elif use_comm:
            sl.comm.gather(sl.frame_local, sl.frame_full, sc.h, sc.w, rdist.BAND_H, 0, sl.stream.cuda_stream)
            sl.last_out = sl.frame_full
        else:
            # torch's collective runs on torch's current stream: order it behind this slot's render
            with torch.cuda.stream(sl.stream):
                sl.last_out = rdist.gather_frame(sl.frame_local, sc.h, sc.w, rdist.BAND_H)
        return sl.last_out

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # (warm-up frames carry the timing flag too: the first timed frame of a context creates its events)
    n_warm = max(args.warmup, n_slots if args.warmup else 0)
    for i in range(n_warm):
        step(i, RT_HIP_TIMING)
    for sl in slots[:min(n_slots, n_warm)]:
        sl.r.kernel_times()                        # discard the warm-up frames' timings
    if use_comm and args.warmup:
        # the two gathers must agree (both outside the timed region)
        fence()
        want = rdist.gather_frame(frame_local, sc.h, sc.w, rdist.BAND_H)
        if rank == 0 and not all(torch.equal(want, sl.frame_full) for sl in slots):
            raise SystemExit("rt_hip_gather_frame and torch.distributed.gather assembled different frames")
    fence()

    # ---- timed region: exactly K steps ------------------------------------
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i, RT_HIP_TIMING)
    fence()
    elapsed = time.perf_counter() - t0
    # per-kernel HIP-event times, averaged over the timed steps (the events are
    # recorded on the stream the kernels run on); read after the timed region
    used = slots[:min(n_slots, args.steps)]
    # wall-clock time per step during which a kernel had at least one launch running, over ALL frames in flight
    # (launch intervals of every slot on one clock; the union never exceeds the timed region)
    spans = {}
    for sl in used:
        for kind, a, b in sl.r.kernel_intervals(origin=used[0].r):
            spans.setdefault(kind, []).append((a, b))
    wall = {k: union_length(v) / args.steps for k, v in spans.items() if k != "frame"}
    per_slot = [sl.r.kernel_times_wall() for sl in used]
    launches = per_slot[0][1]
    kms = {k: sum(p[0][k] for p in per_slot) / len(per_slot) for k in per_slot[0][0]}

    # ---- the timed frames are looked at: every slot's last frame (rendered with the others in flight beside it, on the
    # overlapped schedule and the timed region's grids) against the checker's frame of the same camera, on the device;
    # on a split frame also what rank 0 assembled against the assembled checker frames
    frames_verified = None
    if not args.timed_only:
        ok = True
        ref_cache = {}
        ck = open_checker()
        for sl in used:
            key = sl.last_step if sequence else 0
            if key not in ref_cache:
                ref_cache[key] = checker_frame(ck, camera_of(sl.last_step) if sequence else None)
            ok = ok and torch.equal(sl.frame_local, ref_cache[key])
        ck.close()
        if split:
            ref_full = rdist.gather_frame(ref_cache[0], sc.h, sc.w, rdist.BAND_H)        # (collective)
            if rank == 0:
                ok = ok and all(torch.equal(sl.last_out, ref_full) for sl in used)
        flag = torch.tensor([1 if ok else 0], device=dev)
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        frames_verified = bool(int(flag.item()))
        if not frames_verified:
            raise SystemExit(f"[rank {rank}] a frame of the timed region differs from the frame rendered on its own "
                             f"from the reference's leaf lists (this rank's frames equal: {ok})")
        del ref_cache

    # one frame on its own (nothing else in flight): the latency of a frame, as opposed to the step time
    lat = []
    lone_waves = args.launch_waves if args.launch_waves else 16      # (a frame or a pass alone: every wave slot of a CU)
    r.set_param(RT_PARAM_LAUNCH_WAVES, lone_waves)
    for _ in range(0 if args.timed_only else 5):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        r.render(0, d_out=frame_local.data_ptr())
        r.sync()
        lat.append((time.perf_counter() - t1) * 1e3)
    frame_latency_ms = sorted(lat)[len(lat) // 2] if lat else float("nan")

    # rays of the timed steps (work-list counters; for a sequence every camera is rendered again, untimed)
    if sequence:
        rc = rs = hits = 0
        for i in range(args.steps):
            r.set_camera(camera_of(args.warmup + i))
            r.render(0, d_out=frame_local.data_ptr())
            st = r.stats()
            rc, rs, hits = rc + st["raysClosest"], rs + st["raysShadow"], hits + st["hits"]
        cnt = torch.tensor([rc, rs, hits], dtype=torch.float64, device=dev)
    else:
        st = r.stats()                  # ray counters of the last frame (identical for every frame)
        cnt = torch.tensor([st["raysClosest"], st["raysShadow"], st["hits"]], dtype=torch.float64, device=dev) * args.steps

    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    rays_total = float(cnt[0] + cnt[1])

    # ---- outside the timed region: the same frame with every pass alone on the
    # chip (what a kernel takes when it does not share the GPU with the passes
    # that the normal schedule runs beside it)
    if sequence:
        r.set_camera(camera_of(args.warmup))
    kms_iso = None
    if not args.timed_only:
        r.set_param(RT_PARAM_OVERLAP, 0)
        for _ in range(3):
            r.render(RT_HIP_TIMING, d_out=frame_local.data_ptr())
        r.sync()
        kms_iso, _ = r.kernel_times()
        r.set_param(RT_PARAM_OVERLAP, 0 if args.no_overlap else 1)
    r.set_param(RT_PARAM_LAUNCH_WAVES, waves_for(world if split else 1, n_slots))

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
    dominant = max(("closest", "shadow", "shade"), key=lambda k: wall[k])
    dom_bytes = kbytes[dominant]
    n_launch = max(launches.get(dominant, 1), 1)
    gbs = lambda b, ms: b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    achieved = gbs(dom_bytes, wall[dominant])
    frame_bytes = algorithmic_bytes(sst["nodeVisits"], sst["triTests"], primary, sst["hits"])
    all_rays = max(sst["raysClosest"] + sst["raysShadow"], 1)

    # ---- strong-scaling proxy: what one rank of an N-GPU split of this frame takes on this GPU
    proxy = None
    if world == 1 and not sequence and not args.no_proxy and not args.timed_only:
        proxy = {}
        for parts in (2, 4, 8):
            # (small shares are rendered with more frames in flight: their contexts are only opened now - streams that
            # exist take hardware queues whether they are used or not)
            while len(all_slots) < slots_for(parts):
                add_slot(all_slots)
                torch.cuda.synchronize()
            use = all_slots[:slots_for(parts)]
            outs = []
            for sl in use:
                sl.r.set_param(RT_PARAM_LAUNCH_WAVES, waves_for(parts, len(use)))
                sl.r.set_tile(rdist.BAND_H, parts, 0)
                outs.append(torch.zeros((rdist.padded_rows(sc.h, rdist.BAND_H, parts), sc.w, 4), dtype=torch.uint8, device=dev))
            for i in range(2 * len(use)):
                use[i % len(use)].r.render(0, d_out=outs[i % len(use)].data_ptr())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(40):
                use[i % len(use)].r.render(0, d_out=outs[i % len(use)].data_ptr())
            torch.cuda.synchronize()
            proxy[f"part_0_of_{parts}_ms"] = round((time.perf_counter() - t0) / 40 * 1e3, 4)
            proxy[f"part_0_of_{parts}_frames_in_flight"] = len(use)
            proxy[f"part_0_of_{parts}_waves_per_cu_and_pass"] = waves_for(parts, len(use))
        for sl in all_slots:
            sl.r.set_param(RT_PARAM_LAUNCH_WAVES, waves_for(world if split else 1, n_slots))
            sl.r.set_tile(rdist.BAND_H, 1, 0)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = rays_total / elapsed / 1e6
        if proxy is not None:
            proxy["full_frame_ms"] = round(ms_per_step, 4)
            proxy["note"] = ("step time of rank 0's share of an N-way band split of this frame on ONE GPU, scheduled as the timed "
                             "region of an N-rank run schedules it (frames in flight as given), render only (no exchange): "
                             "full_frame_ms / part_0_of_N_ms bounds the N-GPU speed-up from above")
        counters = committed_counters(args, dominant, world)
        out = {
            "metric": "Mrays/s (primary+shadow) on 1M-tri scene at 4K" if args.workload in ("config3", "config5")
                      else "Mrays/s (primary+shadow)",
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak" if sequence else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # every slot's last timed frame == the same camera's frame rendered alone, passes in order, from the
            # reference's leaf lists (compared on the device after the timed region; a difference ends the run)
            "frames_verified": frames_verified,
            "config": {"workload": f"{args.workload}: {wl['text']}",
                       "kd_tree": {"max_depth": args.depth, "max_prims_in_leaf": args.leaf, "depth": int(tree.depth),
                                   "nodes": int(len(tree.nodes)), "triangle_refs": int(len(tree.recs)),
                                   "triangle_refs_reference_leaves": int(ref_refs),
                                   "leaves": ("the reference's lists" if args.prune_leaves <= 0 else
                                              f"reference nodes; a leaf lists only the triangles that touch its cell grown by "
                                              f"{args.prune_leaves:g} x scene size (same frame, fewer tests)"),
                                   "host_build_s": round(build_s, 3), "gpu_build_s": round(gpu_build_s, 4)},
                       "schedule": "passes in order on one stream" if args.no_overlap else
                                   "closest-hit chain on the main stream, shadow + lighting passes on two auxiliary "
                                   "streams (they overlap the chain)",
                       "rays_per_step": int(rays_total / args.steps), "closest_rays_per_step": int(cnt[0] / args.steps),
                       "shadow_rays_per_step": int(cnt[1] / args.steps),
                       "frames_per_s": round(args.steps * (world if sequence else 1) / elapsed, 3),
                       "frames_in_flight": n_slots, "warmup_frames_rendered": n_warm, "frame_latency_ms_alone": round(frame_latency_ms, 4) if lat else None,
                       "waves_per_cu_and_pass": {"timed_region": waves_for(world if split else 1, n_slots),
                                                 "frame_alone_and_isolated_passes": lone_waves},
                       "parallelism": "single GPU" if world == 1 else
                                      (f"frame f of the sequence on rank f mod {world}, scene replicated, no exchange" if sequence else
                                       f"frame split in bands of {rdist.BAND_H} rows over {world} GPUs, scene replicated, "
                                       f"rows gathered on rank 0: {gather_kind}")},
            # sums of launch durations (launches that run side by side add up: can exceed the frame time)
            "kernel_ms_per_frame_rank0": {k: round(x, 4) for k, x in kms.items()},
            # wall-clock time per step during which at least one launch of the kernel ran, all frames in flight
            # taken together (never exceeds ms_per_step)
            "kernel_wall_ms_per_frame_rank0": {k: round(x, 4) for k, x in wall.items()},
            "kernel_ms_per_frame_isolated_rank0": {k: round(x, 4) for k, x in kms_iso.items()} if kms_iso else None,
            "roofline": {"bound": "hbm", "kernel": "k_" + dominant,
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": counters.get("fabric_bytes_per_launch") if counters else None,
                         "time_basis": "wall-attributed: algorithmic bytes of the kernel's launches of one frame / the time per step "
                                       "during which at least one launch of the kernel ran, over all frames in flight (HIP events "
                                       "on the streams they run on, timed region, one clock)",
                         "launches_per_frame": n_launch,
                         "algorithmic_bytes_per_launch": int(dom_bytes / n_launch),
                         "avg_launch_ms": round(kms[dominant] / n_launch, 4),
                         "avg_launch_ms_wall": round(wall[dominant] / n_launch, 4),
                         "avg_launch_ms_isolated": round(kms_iso[dominant] / n_launch, 4) if kms_iso else None,
                         "achieved_isolated": round(gbs(dom_bytes, kms_iso[dominant]), 2) if kms_iso else None,
                         # the working set (0.26 GB) is served by L2 and the Infinity Cache, so these are the roofs that can bind
                         "frac_of_l2_roof": round(achieved / L2_PEAK_GBS, 5), "l2_peak": L2_PEAK_GBS,
                         "frac_of_infinity_cache_roof": round(achieved / MALL_PEAK_GBS, 5), "infinity_cache_peak": MALL_PEAK_GBS,
                         "frame_algorithmic_bytes": int(frame_bytes),
                         # all kernels of a frame: algorithmic bytes of this rank's frame / the step time
                         "frame_achieved_GBs": round(gbs(frame_bytes, ms_per_step), 2),
                         "per_ray": {"nodes": round(sst["nodeVisits"] / all_rays, 2), "tri_tests": round(sst["triTests"] / all_rays, 2)},
                         "traversal_stack": {"deepest": int(sst["maxStack"]), "pushes_beyond_lds": int(sst["spillPushes"]),
                                             "rays": int(all_rays)},
                         "counters": counters,
                         # which committed profiler file backs which figure (written by tools/profile_round.sh on the same frame)
                         "evidence": {
                             "achieved_isolated / avg_launch_ms_isolated": PROFILE_TAG + "_kernel_stats_serial.csv: rocprofv3 --kernel-trace "
                                 "--stats of `bench.py --timed-only --no-overlap --in-flight 1 --launch-waves 16` (one frame at a "
                                 "time, passes in order): algorithmic_bytes_per_launch / AverageNs of the kernel",
                             "avg_launch_ms": PROFILE_TAG + "_kernel_stats.csv: the same of `bench.py --timed-only` (launches of the frames in "
                                 "flight overlap, so these durations add up to more than the step time)",
                             "achieved / frac": "HIP events of this run (kernel_wall_ms_per_frame_rank0); no profiler file can back a "
                                 "wall-attributed time; ms_per_step bounds it from above",
                             "traffic / counters": PROFILE_TAG + "_counters.json (rocprofv3 --pmc passes, tools/pmc_passes.sh), not measured in this run"}},
        }
        if proxy is not None:
            out["strong_scaling_proxy"] = proxy
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc, v, t, tree, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
    for sl in all_slots:
        if sl.comm:
            sl.comm.close()
    for sl in reversed(all_slots):               # (the first one owns the scene the others read)
        sl.r.close()

    # ---- the other two single-GPU workloads, a few steps each (so that whoever times this command times them as well);
    # after this workload's contexts have been closed: their streams would share hardware queues with the new ones
    other = None
    if world == 1 and args.workload == "config3" and not args.timed_only and not args.no_other_configs:
        # (a 1080p frame takes 0.8 ms: 5 steps would time little more than the filling and draining of four contexts)
        other = {"config2": side_workload("config2", local_dev, dev, 22, args.leaf, args.prune_leaves, 20),
                 # (the fly-through renders the geometry of this workload: same trees)
                 "config5": side_workload("config5", local_dev, dev, args.depth, args.leaf, args.prune_leaves, 8, tree=tree, plain=plain)}
        if not all(o["frames_verified"] for o in other.values()):
            raise SystemExit("a frame of one of the other workloads differs from the frame rendered on its own: "
                             + json.dumps({k: o["frames_verified"] for k, o in other.items()}))

    if plain is not None:
        plain.release()
    if rank == 0:
        if other is not None:
            out["other_configs"] = other
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
