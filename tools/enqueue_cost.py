"""Host time of enqueueing one frame (rt_hip_render returns when the frame is queued):
python tools/enqueue_cost.py [config3|config2] [parts]"""
import os, sys, time
os.environ["GPU_MAX_HW_QUEUES"] = "16"
from raytracing_amd import scenes
from raytracing_amd.pipe import HipRenderer, FlatTree, RT_HIP_TIMING

name = sys.argv[1] if len(sys.argv) > 1 else "config3"
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sc = getattr(scenes, name)()
v, t = sc.flatten()
tree = FlatTree.build(v, t, 26 if name != "config2" else 22, 12, prune_leaves=1e-5)
rs = [HipRenderer(0) for _ in range(3)]
rs[0].load_scene(sc, tree)
for r in rs[1:]:
    r.share_scene(rs[0], sc)
for r in rs:
    r.set_tile(16, parts, 0)
    r.render(0)
for r in rs:
    r.sync()
for flags in (0, RT_HIP_TIMING):
    n = 60
    t0 = time.perf_counter()
    for i in range(n):
        rs[i % 3].render(flags)
    t1 = time.perf_counter()
    for r in rs:
        r.sync()
    t2 = time.perf_counter()
    print(f"{name} 1/{parts} frame, flags {flags}: enqueue {1e3 * (t1 - t0) / n:.3f} ms per frame, step {1e3 * (t2 - t0) / n:.3f} ms", flush=True)
for r in reversed(rs):
    r.close()
