"""Host vs GPU kd-tree build time for the benchmark scenes (GPU: from ordinary and from page-locked input arrays)."""
import sys, time
import torch
from raytracing_amd import scenes
from raytracing_amd.pipe import DeviceTree, FlatTree


def pinned(a):
    """The same structured array in page-locked memory."""
    t = torch.empty(a.nbytes, dtype=torch.uint8).pin_memory()
    out = t.numpy().view(a.dtype)
    out[:] = a
    return out, t


for name, depth, leaf in (("config2", 24, 12), ("config3", 26, 12), ("config3", 10, 7), ("default_scene", 10, 7), ("default_scene", 14, 7)):
    sc = getattr(scenes, name)()
    v, t = sc.flatten()
    t0 = time.perf_counter(); h = FlatTree.build(v, t, depth, leaf); th = time.perf_counter() - t0
    res = []
    for label in ("", " from page-locked arrays"):
        vv, tt, keep = v, t, None
        if label:
            (vv, k1), (tt, k2) = pinned(v), pinned(t)
            keep = (k1, k2)
        DeviceTree.build(vv, tt, depth, leaf).release()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); d = DeviceTree.build(vv, tt, depth, leaf); best = min(best, time.perf_counter() - t0)
            n = (d.n_nodes, d.n_recs); d.release()
        res.append(f"gpu{label} {best*1e3:.2f} ms")
    print(f"{name} {len(t)} tris ({len(v)} vertices) depth {depth} leaf {leaf}: host {th*1e3:.1f} ms, {', '.join(res)}, nodes {n[0]} refs {n[1]} (host {len(h.nodes)} {len(h.recs)})", flush=True)
    h.release()
