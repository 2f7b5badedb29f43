"""One frame at a time, rendered as K band-interleaved parts on K contexts side by side (the frames-in-flight
machinery applied inside a frame): latency of the whole frame against one context rendering it alone.
python tools/split_frame.py [config3|config5|config2]"""
import os, sys, time
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("QUEUES", "16")
import torch
from raytracing_amd import scenes
from raytracing_amd.pipe import HipRenderer, FlatTree, RT_PARAM_LAUNCH_WAVES

name = sys.argv[1] if len(sys.argv) > 1 else "config3"
sc = getattr(scenes, name)()
v, t = sc.flatten()
tree = FlatTree.build(v, t, 22 if name == "config2" else 26, 12, prune_leaves=1e-5)
KMAX = 4
streams = [torch.cuda.Stream() for _ in range(KMAX)]
rs = []
for k in range(KMAX):
    r = HipRenderer(0, stream=streams[k].cuda_stream)
    if k:
        r.share_scene(rs[0], sc)
    else:
        r.load_scene(sc, tree)
    rs.append(r)
outs = [torch.zeros((sc.h, sc.w, 4), dtype=torch.uint8, device="cuda") for _ in range(KMAX)]
for K in (1, 2, 3, 4):
    for waves in (8, 12, 16, 20):
        for k in range(K):
            rs[k].set_tile(16, K, k)
            rs[k].set_param(RT_PARAM_LAUNCH_WAVES, waves)
        best = 1e9
        for rep in range(8):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(K):
                rs[k].render(0, d_out=outs[k].data_ptr())
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print(f"{name}: frame as {K} part(s) side by side, {waves} waves per CU and pass: {best * 1e3:.3f} ms", flush=True)
for r in reversed(rs):
    r.close()
