#!/usr/bin/env python3
"""Fly-through of the benchmark scene: 240 cameras on a circle around the
sphere, reflective + refractive material; frame f is rendered by rank
f mod N (scene and tree replicated, no data-path collective) and written as
<out>/<f + 8>.png, the numbering the reference's build_video.sh expects
(avconv -start_number 8 -i pngs/%d.png).

    python apps/flythrough.py --frames 240 --out pngs
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 apps/flythrough.py --frames 240

Prints one JSON line per rank with its frame count and frames per second.
"""
import argparse
import json
import os
import struct
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

from raytracing_amd import dist as rdist
from raytracing_amd import scenes
from raytracing_amd.pipe import FlatTree, HipRenderer


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
    args = ap.parse_args()
    w, h = (int(x) for x in args.size.split("x"))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    sc = scenes.config5_frame(0, w, h, args.lon, args.lat)
    v, t = sc.flatten()
    tree = FlatTree.build(v, t, args.depth, args.leaf)
    r = HipRenderer(local_rank)
    r.load_scene(sc, tree)                       # geometry and tree stay resident for the whole sequence
    if not args.no_files:
        os.makedirs(args.out, exist_ok=True)
    mine = rdist.sequence_frames(args.frames, world, rank)
    t0 = time.perf_counter()
    for f in mine:
        r.set_camera(scenes.make_camera(w, h, world=scenes.orbit_world(f, args.frames)))
        r.render(0)
        img = r.read_frame()
        if not args.no_files:
            write_png(os.path.join(args.out, f"{f + 8}.png"), img)
    dt = time.perf_counter() - t0
    st = r.stats()
    r.close()
    print(json.dumps({"rank": rank, "world": world, "frames": len(mine), "seconds": round(dt, 3),
                      "frames_per_s": round(len(mine) / dt, 3) if mine else 0.0,
                      "rays_last_frame": st["raysClosest"] + st["raysShadow"]}), flush=True)


if __name__ == "__main__":
    main()
