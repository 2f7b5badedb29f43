"""Development helper for profiler passes: renders the benchmark frame a few
times without the CPU baseline or torch (short process, kernels only)."""
import os, sys
from raytracing_amd import scenes
from raytracing_amd.pipe import HipRenderer, FlatTree, RT_PARAM_OVERLAP

name = sys.argv[1] if len(sys.argv) > 1 else "config3"
depth, leaf, frames = (int(x) for x in (sys.argv[2:5] + ["28", "16", "3"][len(sys.argv[2:5]):]))
sc = getattr(scenes, name)()
v, t = sc.flatten()
tree = FlatTree.build(v, t, depth, leaf, prune_leaves=float(os.environ.get("RT_PRUNE", "1e-5")))    # as bench.py builds it
r = HipRenderer(0)
r.load_scene(sc, tree)
if os.environ.get("RT_OVERLAP") is not None:
    r.set_param(RT_PARAM_OVERLAP, int(os.environ["RT_OVERLAP"]))
for _ in range(frames):
    r.render(0)
r.sync()
print("rendered", frames, "frames", r.stats())
r.close()
