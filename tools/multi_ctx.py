"""Experiment: K contexts on one GPU, each renders every K-th band of the frame on its own stream."""
import os, sys, time
import numpy as np
import torch
from raytracing_amd import scenes
from raytracing_amd.pipe import HipRenderer, FlatTree, RT_PARAM_OVERLAP

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ov = int(sys.argv[2]) if len(sys.argv) > 2 else 1
E = int(sys.argv[3]) if len(sys.argv) > 3 else 1        # external parts (this GPU renders part 0 of E)
sc = scenes.config3()
v, t = sc.flatten()
tree = FlatTree.build(v, t, 26, 12)
streams = [torch.cuda.Stream() for _ in range(K)]
rs = []
for k in range(K):
    r = HipRenderer(0, stream=streams[k].cuda_stream)
    r.load_scene(sc, tree)
    r.set_param(RT_PARAM_OVERLAP, ov)
    r.set_tile(16, K * E, k * E)
    rs.append(r)
for it in range(3):
    for r in rs: r.render(0)
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    t0 = time.perf_counter()
    for it in range(10):
        for r in rs: r.render(0)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 10)
print(f"K={K} overlap={ov} E={E}: {best*1e3:.2f} ms per frame")
for r in rs: r.close()
