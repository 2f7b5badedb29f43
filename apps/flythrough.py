#!/usr/bin/env python3
"""Fly-through of the benchmark scene: 240 cameras on a circle around the
sphere, reflective + refractive material; frame f is rendered by rank
f mod N (scene and tree replicated, no data-path collective) and written as
<out>/<f + 8>.png, the numbering the reference's build_video.sh expects
(avconv -start_number 8 -i pngs/%d.png).

    python apps/flythrough.py --frames 240 --out pngs
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 apps/flythrough.py --frames 240

Three frames are in flight on the GPU (contexts that share the resident
scene, one host thread each: render, then read back into a page-locked block),
and the PNG encoding of finished frames runs on a pool of encoder threads
beside the next frames' rendering (zlib releases the interpreter lock).
--serial renders, reads back and encodes one frame after the other on one
thread instead.

Prints one JSON line per rank: frames per second end to end (render +
read-back + encoding + file), and of the render + read-back stage alone.
"""
import argparse
import concurrent.futures as cf
import json
import os
import queue
import struct
import sys
import threading
import time
import zlib

# two frames in flight = two contexts of three streams each, plus the null stream: more than the runtime's
# default of four hardware queues per process, which would fold streams together and serialise them
os.environ["GPU_MAX_HW_QUEUES"] = "16"           # before the HIP runtime starts; the machine may export 4

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

from raytracing_amd import dist as rdist
from raytracing_amd import scenes
from raytracing_amd.pipe import RT_PARAM_LAUNCH_WAVES, FlatTree, HipRenderer


def write_png(path, bgra):
    h, w, _ = bgra.shape
    rows = np.zeros((h, 1 + 3 * w), np.uint8)
    rows[:, 1:] = bgra[..., [2, 1, 0]].reshape(h, 3 * w)

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data))
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(rows.tobytes(), 1)) + chunk(b"IEND", b""))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=240)
    ap.add_argument("--size", default="3840x2160")
    ap.add_argument("--lon", type=int, default=1000)
    ap.add_argument("--lat", type=int, default=500)
    ap.add_argument("--depth", type=int, default=26)
    ap.add_argument("--leaf", type=int, default=12)
    ap.add_argument("--out", default="pngs")
    ap.add_argument("--no-files", action="store_true")
    ap.add_argument("--serial", action="store_true", help="one frame at a time: render, read back, encode")
    ap.add_argument("--in-flight", type=int, default=3, help="frames rendered side by side on the GPU")
    ap.add_argument("--prune-leaves", type=float, default=0.0,
                    help="margin (x scene size) of the host builder's leaf pruning, e.g. 1e-5; 0 = the reference's leaf lists "
                         "(pruned lists are not checked against them here: compare a frame of both once, as bench.py does)")
    ap.add_argument("--encoders", type=int, default=0, help="PNG encoder threads (default: host cores of this rank, at most 12)")
    args = ap.parse_args()
    w, h = (int(x) for x in args.size.split("x"))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_slots = 1 if args.serial else max(1, args.in_flight)
    n_enc = args.encoders or max(1, min(12, (os.cpu_count() or 1) // max(world, 1)))

    sc = scenes.config5_frame(0, w, h, args.lon, args.lat)
    v, t = sc.flatten()
    tree = FlatTree.build(v, t, args.depth, args.leaf, prune_leaves=args.prune_leaves)
    rs = [HipRenderer(local_rank)]
    rs[0].load_scene(sc, tree)                   # geometry and tree stay resident for the whole sequence
    for _ in range(1, n_slots):
        r = HipRenderer(local_rank)
        r.share_scene(rs[0], sc)                 # no second copy of the scene
        rs.append(r)
    if n_slots > 1:
        for r in rs:                             # frames in flight: a pass leaves half of every CU to the other frames' passes
            r.set_param(RT_PARAM_LAUNCH_WAVES, 8)
    if not args.no_files:
        os.makedirs(args.out, exist_ok=True)
    mine = rdist.sequence_frames(args.frames, world, rank)

    def camera(f):
        return scenes.make_camera(w, h, world=scenes.orbit_world(f, args.frames))

    def encode(f, img):
        if not args.no_files:
            write_png(os.path.join(args.out, f"{f + 8}.png"), img)

    gpu_busy = [0.0] * n_slots                   # per slot: time spent in render + read-back
    t0 = time.perf_counter()
    if args.serial:
        r = rs[0]
        for f in mine:
            t1 = time.perf_counter()
            r.set_camera(camera(f))
            r.render(0)
            img = r.read_frame()
            gpu_busy[0] += time.perf_counter() - t1
            encode(f, img)
    else:
        # page-locked blocks travel in a ring: a render thread takes a free one, fills it, hands it to an encoder,
        # which returns it when the file is written
        blocks = queue.Queue()
        owner = {}
        for k in range(n_slots + n_enc):
            try:
                a = rs[0].alloc_host_frame()
                owner[id(a)] = a
            except Exception:                     # the driver refused: pageable memory works too (slower copy)
                a = np.zeros((h, w, 4), np.uint8)
            blocks.put(a)
        errs = []

        def finish(f, a):
            try:
                encode(f, a)
            except Exception as e:               # noqa: BLE001 - reported by the main thread
                errs.append(repr(e))
            finally:
                blocks.put(a)

        with cf.ThreadPoolExecutor(n_enc) as pool:
            def render_thread(s):
                try:
                    r = rs[s]
                    for f in mine[s::n_slots]:
                        a = blocks.get()
                        t1 = time.perf_counter()
                        r.set_camera(camera(f))
                        r.render(0)
                        r.read_frame(out=a)      # waits for this context's frame only; the other slot keeps the GPU busy
                        gpu_busy[s] += time.perf_counter() - t1
                        pool.submit(finish, f, a)
                except Exception as e:           # noqa: BLE001
                    errs.append(repr(e))
            th = [threading.Thread(target=render_thread, args=(s,)) for s in range(n_slots)]
            for x in th:
                x.start()
            for x in th:
                x.join()
        if errs:
            raise SystemExit("fly-through failed: " + "; ".join(errs))
        for a in owner.values():
            rs[0].free_host_frame(a)
    dt = time.perf_counter() - t0
    st = rs[0].stats()
    for r in reversed(rs):
        r.close()
    stage = max(gpu_busy) if not args.serial else gpu_busy[0]
    print(json.dumps({"rank": rank, "world": world, "frames": len(mine), "seconds": round(dt, 3),
                      "frames_per_s": round(len(mine) / dt, 3) if mine else 0.0,
                      "render_readback_frames_per_s": round(len(mine) / stage, 3) if mine and stage > 0 else 0.0,
                      "frames_in_flight": n_slots, "encoder_threads": 0 if args.serial else n_enc,
                      "files": not args.no_files,
                      "rays_last_frame": st["raysClosest"] + st["raysShadow"]}), flush=True)


if __name__ == "__main__":
    main()
