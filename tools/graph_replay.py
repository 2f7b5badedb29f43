"""Experiment: three frames in flight, each context's frame captured once in a HIP graph and replayed, against
the same frames enqueued call by call.  python tools/graph_replay.py [config3|config2] [parts]"""
import os, sys, time
os.environ["GPU_MAX_HW_QUEUES"] = "16"
import torch
from raytracing_amd import scenes
from raytracing_amd.pipe import HipRenderer, FlatTree, RT_PARAM_LAUNCH_WAVES

name = sys.argv[1] if len(sys.argv) > 1 else "config2"
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sc = getattr(scenes, name)()
v, t = sc.flatten()
tree = FlatTree.build(v, t, 26 if name != "config2" else 22, 12, prune_leaves=1e-5)
K = 3
streams = [torch.cuda.Stream() for _ in range(K)]
rs = [HipRenderer(0, stream=s.cuda_stream) for s in streams]
rs[0].load_scene(sc, tree)
for r in rs[1:]:
    r.share_scene(rs[0], sc)
outs = []
for r in rs:
    r.set_param(RT_PARAM_LAUNCH_WAVES, 8 if parts <= 2 else 6)
    r.set_tile(16, parts, 0)
    outs.append(torch.zeros((sc.h, sc.w, 4), dtype=torch.uint8, device="cuda"))
for k in range(2 * K):
    rs[k % K].render(0, d_out=outs[k % K].data_ptr())
torch.cuda.synchronize()
n = 90
t0 = time.perf_counter()
for i in range(n):
    rs[i % K].render(0, d_out=outs[i % K].data_ptr())
torch.cuda.synchronize()
direct = (time.perf_counter() - t0) / n
want = [o.clone() for o in outs]
graphs = []
for k in range(K):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=streams[k]):
        rs[k].render(0, d_out=outs[k].data_ptr())
    graphs.append(g)
for o in outs:
    o.zero_()
for k in range(2 * K):
    graphs[k % K].replay()
torch.cuda.synchronize()
same = all(torch.equal(a, b) for a, b in zip(outs, want))
t0 = time.perf_counter()
for i in range(n):
    graphs[i % K].replay()
torch.cuda.synchronize()
replay = (time.perf_counter() - t0) / n
print(f"{name} 1/{parts} frame, three in flight: {direct * 1e3:.3f} ms per frame call by call, {replay * 1e3:.3f} ms replayed from graphs, same frames {same}", flush=True)
del graphs
for r in reversed(rs):
    r.close()
