"""Host vs GPU kd-tree build time for the benchmark scenes."""
import sys, time
from raytracing_amd import scenes
from raytracing_amd.pipe import DeviceTree, FlatTree

for name, depth, leaf in (("config2", 24, 12), ("config3", 26, 12), ("config3", 10, 7), ("default_scene", 10, 7), ("default_scene", 14, 7)):
    sc = getattr(scenes, name)()
    v, t = sc.flatten()
    t0 = time.perf_counter(); h = FlatTree.build(v, t, depth, leaf); th = time.perf_counter() - t0
    DeviceTree.build(v, t, depth, leaf).release()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); d = DeviceTree.build(v, t, depth, leaf); best = min(best, time.perf_counter() - t0)
        n = (d.n_nodes, d.n_recs); d.release()
    print(f"{name} {len(t)} tris depth {depth} leaf {leaf}: host {th*1e3:.1f} ms, gpu {best*1e3:.1f} ms, nodes {n[0]} refs {n[1]} (host {len(h.nodes)} {len(h.recs)})", flush=True)
    h.release()
