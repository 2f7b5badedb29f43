"""Work counters of one frame of a benchmark workload (counting build of the kernels):
python tools/frame_stats.py [config3|config2|config5] [depth] [leaf]"""
import json, sys
from raytracing_amd import scenes
from raytracing_amd.pipe import HipRenderer, FlatTree, RT_HIP_STATS

name = sys.argv[1] if len(sys.argv) > 1 else "config3"
depth, leaf = (int(x) for x in (sys.argv[2:4] + ["26", "12"][len(sys.argv[2:4]):]))
sc = getattr(scenes, name)()
v, t = sc.flatten()
tree = FlatTree.build(v, t, depth, leaf)
r = HipRenderer(0)
r.load_scene(sc, tree)
r.render(RT_HIP_STATS)
st = r.stats()
r.close()
rays = st["raysClosest"] + st["raysShadow"]
st["waveDescendIterations"] = st["nodeSlots"] / 64
st["waveLeafIterations"] = st["triSlots"] / 64
st["laneUseDescend"] = (st["nodeVisits"]) / max(st["nodeSlots"], 1)
st["laneUseLeaf"] = st["triTests"] / max(st["triSlots"], 1)
st["perRay"] = {"nodes": st["nodeVisits"] / rays, "tris": st["triTests"] / rays}
print(json.dumps(st, indent=1))
