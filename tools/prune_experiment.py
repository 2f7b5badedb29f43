"""Experiment: leaf references whose triangle does not touch the leaf's box are dropped
(separating-axis test with a safety margin), the nodes stay the reference's.  Prints the
share of references dropped, renders the frame with both trees on the GPU, compares the
images and primary hit ids, and times both (best of a few frames, one frame at a time).

  python tools/prune_experiment.py [config3|config2|config5] [depth] [leaf] [margin]
"""
import sys
import time

import numpy as np

from raytracing_amd import scenes
from raytracing_amd.pipe import RT_HIP_HITIDS, RT_HIP_STATS, RT_HIP_TIMING, FlatTree, HipRenderer


def node_boxes(nodes, bbox):
    n = len(nodes)
    lo = np.zeros((n, 3))
    hi = np.zeros((n, 3))
    c = np.array([bbox["center"][k] for k in "xyz"], np.float64)
    e = np.array([bbox["extents"][k] for k in "xyz"], np.float64)
    lo[0], hi[0] = c - e, c + e
    last = nodes["isLast"]
    left, right, sep, axis = nodes["leftNode"], nodes["rightNode"], nodes["sep"], nodes["axis"]
    for i in range(n):                     # parents come before their children
        if last[i]:
            continue
        a, l, r = axis[i], left[i], right[i]
        lo[l], hi[l] = lo[i], hi[i].copy()
        hi[l][a] = sep[i]
        lo[r], hi[r] = lo[i].copy(), hi[i]
        lo[r][a] = sep[i]
    return lo, hi


def touches(v0, v1, v2, bc, bh):
    """Triangle / box overlap (separating axes: 3 box normals, triangle normal, 9 edge cross products)."""
    a0, a1, a2 = v0 - bc, v1 - bc, v2 - bc
    ok = np.ones(len(bc), bool)
    for k in range(3):
        mn = np.minimum(np.minimum(a0[:, k], a1[:, k]), a2[:, k])
        mx = np.maximum(np.maximum(a0[:, k], a1[:, k]), a2[:, k])
        ok &= ~((mn > bh[:, k]) | (mx < -bh[:, k]))
    nrm = np.cross(a1 - a0, a2 - a0)
    ok &= np.abs((nrm * a0).sum(1)) <= (np.abs(nrm) * bh).sum(1)
    for f in (a1 - a0, a2 - a1, a0 - a2):
        for k in range(3):
            u = np.zeros(3)
            u[k] = 1.0
            ax = np.cross(np.broadcast_to(u, f.shape), f)
            p0, p1, p2 = (ax * a0).sum(1), (ax * a1).sum(1), (ax * a2).sum(1)
            rr = (np.abs(ax) * bh).sum(1)
            ok &= ~((np.minimum(np.minimum(p0, p1), p2) > rr) | (np.maximum(np.maximum(p0, p1), p2) < -rr))
    return ok


def pruned(tree, verts, tris, margin):
    nodes, idx = tree.expand()
    lo, hi = node_boxes(nodes, tree.bbox)
    leaf = np.nonzero(nodes["isLast"] != 0)[0]
    cnt = nodes["primsCount"][leaf].astype(np.int64)
    first = nodes["prims"][leaf].astype(np.int64)
    li = np.repeat(leaf, cnt)
    ri = np.repeat(first, cnt) + (np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt))
    tri = idx[ri].astype(np.int64)
    P = np.stack([verts["pos"][k] for k in "xyz"], 1).astype(np.float64)
    v0, v1, v2 = P[tris["pV0"][tri]], P[tris["pV1"][tri]], P[tris["pV2"][tri]]
    bc, bh = (lo[li] + hi[li]) / 2, (hi[li] - lo[li]) / 2
    size = np.abs(bc).max(1, keepdims=True) + bh.max(1, keepdims=True)
    keep = touches(v0, v1, v2, bc, bh + margin * size)
    new_cnt = np.bincount(li[keep], minlength=len(nodes))[leaf]
    out = nodes.copy()
    out["primsCount"][leaf] = new_cnt
    out["prims"][leaf] = np.cumsum(new_cnt) - new_cnt
    return out, idx[ri[keep]], keep.mean()


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "config3"
    depth, leaf = (int(x) for x in (sys.argv[2:4] + ["26", "12"][len(sys.argv[2:4]):]))
    margin = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-5
    sc = getattr(scenes, name)()
    v, t = sc.flatten()
    tree = FlatTree.build(v, t, depth, leaf)
    t0 = time.time()
    pn, pi, kept = pruned(tree, v, t, margin)
    print(f"{name} depth {depth} leaf {leaf}: {len(tree.recs)} references, kept {kept:.4f} ({time.time() - t0:.1f} s)", flush=True)
    small = FlatTree.from_interchange(pn, pi, v, t, tree.bbox)
    r = HipRenderer(0)
    res = {}
    for label, tr in (("reference leaves", tree), ("pruned leaves", small), ("reference leaves again", tree)):
        r.load_scene(sc, tr)
        best = None
        for _ in range(5):
            r.render(RT_HIP_TIMING)
            ms, _ = r.kernel_times()
            if best is None or ms["frame"] < best["frame"]:
                best = ms
        r.render(RT_HIP_HITIDS | RT_HIP_STATS)
        img, ids, st = r.read_frame().copy(), r.read_hit_ids().copy(), r.stats()
        res[label] = (img, ids)
        rays = st["raysClosest"] + st["raysShadow"]
        print(f"{label}: frame {best['frame']:.3f} ms (closest {best['closest']:.2f} shadow {best['shadow']:.2f}), "
              f"{st['nodeVisits'] / rays:.1f} nodes + {st['triTests'] / rays:.1f} tests per ray", flush=True)
    a, b = res["reference leaves"], res["pruned leaves"]
    print("pixels that differ:", int((a[0] != b[0]).any(-1).sum()), " hit ids that differ:", int((a[1] != b[1]).sum()), flush=True)
    r.close()


main()
