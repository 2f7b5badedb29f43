"""Development helper: the 8K test scene at several sizes against the oracle, listing pixels that differ."""
import sys
import numpy as np
from raytracing_amd import scenes
from raytracing_amd.pipe import HipRenderer, FlatTree, RT_HIP_STATS, RT_HIP_HITIDS
from oracle import orc

w, h, step = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
flags = int(sys.argv[4]) if len(sys.argv) > 4 else (RT_HIP_STATS | RT_HIP_HITIDS)
import os
if os.environ.get("SCENE") == "config3":
    sc = scenes.config3(w, h)
else:
    sc = scenes.sphere_scene(250, 200, w, h, reflect=0.5, stack_size=5)
v, t = sc.flatten()
tree = FlatTree.build(v, t, int(os.environ.get("DEPTH", "24")), int(os.environ.get("LEAF", "12")))
nodes, idx = tree.expand()
import os
for rep in range(int(os.environ.get('REPS', '2'))):
    r = HipRenderer(0)
    r.load_scene(sc, tree)
    import os
    if os.environ.get('RT_OVERLAP') is not None:
        from raytracing_amd.pipe import RT_PARAM_OVERLAP
        r.set_param(RT_PARAM_OVERLAP, int(os.environ['RT_OVERLAP']))
    r.render(flags) if os.environ.get('WARM') else None
    r.render(flags)
    img, ids, st = r.read_frame(), r.read_hit_ids(), r.stats()
    oimg, oid, ost = orc.raytrace(sc, tree.bbox, nodes, idx, x0=17 % step, x_step=step, y0=5 % step, y_step=step)
    m = np.zeros(ids.shape, bool)
    m[5 % step::step, 17 % step::step] = True
    d = np.abs(img.astype(int) - oimg.astype(int)).max(axis=-1)
    bad = np.argwhere(m & (d > 0))
    print(f"{w}x{h} step {step} rep {rep}: {m.sum()} sampled, {len(bad)} differ (max {d[m].max()}), ids differ {(m & (ids != oid)).sum()}; maxStack {st['maxStack']} spills {st['spillPushes']}", flush=True)
    r.close()
    for y, x in bad[:int(os.environ.get('SHOW', '12'))]:
        print("   ", y, x, img[y, x].tolist(), oimg[y, x].tolist(), ids[y, x], oid[y, x])

