"""Development helper: time the benchmark frame for several kd-tree limits."""
import os, sys, time
import numpy as np
from raytracing_amd import scenes
from raytracing_amd.pipe import HipRenderer, FlatTree, RT_HIP_STATS, RT_HIP_TIMING, RT_PARAM_OVERLAP

def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "config3"
    combos = [tuple(float(x) for x in a.split(",")) for a in sys.argv[2:]] or [(24, 7), (28, 7), (32, 7), (32, 16), (36, 12)]
    sc = getattr(scenes, name)()
    v, t = sc.flatten()
    r = HipRenderer(0)
    for combo in combos:
        depth, leaf = int(combo[0]), int(combo[1])
        stop = combo[2] if len(combo) > 2 else 0.0
        axes = int(combo[3]) if len(combo) > 3 else 0
        t0 = time.time()
        tree = FlatTree.build(v, t, depth, leaf, 0, stop, axes, float(os.environ.get("RT_PRUNE", "1e-5")))
        tb = time.time() - t0
        r.load_scene(sc, tree)
        if os.environ.get("RT_PARTS"):
            r.set_tile(int(os.environ.get("RT_BAND", "16")), int(os.environ["RT_PARTS"]), int(os.environ.get("RT_PART", "0")))
        if os.environ.get("RT_OVERLAP") is not None:
            r.set_param(RT_PARAM_OVERLAP, int(os.environ["RT_OVERLAP"]))
        best = None
        for _ in range(4):
            r.render(RT_HIP_TIMING)
            ms, _ = r.kernel_times()
            if best is None or ms["frame"] < best["frame"]:
                best = ms
        st0 = r.stats()
        r.render(RT_HIP_STATS)
        st = r.stats()
        rays = st0["raysClosest"] + st0["raysShadow"]
        nc, ns = st["nodeVisits"] - st["nodeVisitsShadow"], st["nodeVisitsShadow"]
        tc, ts = st["triTests"] - st["triTestsShadow"], st["triTestsShadow"]
        print(f"{name} depth {depth:2d} leaf {leaf:2d} stop {stop} axes {axes}: build {tb:.2f}s nodes {len(tree.nodes)/1e6:.2f}M refs {len(tree.recs)/1e6:.2f}M "
              f"| frame {best['frame']:.2f} ms (closest {best['closest']:.2f} shadow {best['shadow']:.2f} shade {best['shade']:.2f}) {rays/best['frame']/1e3:.0f} Mrays/s "
              f"| closest/ray: {nc/st['raysClosest']:.1f} nodes {tc/st['raysClosest']:.1f} tris; shadow/ray: {ns/max(st['raysShadow'],1):.1f} nodes {ts/max(st['raysShadow'],1):.1f} tris "
              f"| maxStack {st['maxStack']} spills {st['spillPushes']} "
              f"| lane use closest: descend {nc/max(st['nodeSlots'],1):.2f} leaf {tc/max(st['triSlots'],1):.2f}; "
              f"shadow: descend {ns/max(st['nodeSlotsShadow'],1):.2f} leaf {ts/max(st['triSlotsShadow'],1):.2f}", flush=True)
        print("   slot rays", st["slotRays"][:sc.stack_size], "hits", st["slotHits"][:sc.stack_size], flush=True)
        tree.release()
    r.close()

main()
