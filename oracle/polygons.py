"""CPU restatement of "label raster -> polygon rings" (TEST INFRASTRUCTURE ONLY, like everything under oracle/).

The reference vectorises with rasterio.features.shapes (GDAL polygonize, 4-connected) per segment id
(obia/segmentation/segment_boundaries.py:59-71).  GDAL / rasterio are not installed here, so this restatement is pinned
only by hand-made known answers and by properties (ring area == pixel count, rasterising the rings back gives the
label map): **parity unpinned** with respect to GDAL's vertex order and ring orientation; the geometry (the point set of
every polygon) is what the properties fix.

Algorithm (dictionary stitching, deliberately different in structure from the one-lane-per-ring walk of
obia_amd/csrc/polygons.hip): every pixel side that separates label L from anything else becomes a directed edge with L
on its right (top side heads east, right side south, bottom side west, left side north); at a corner the next edge is
the right turn if it exists, else straight, else the left turn (keeps diagonal-only contacts apart = 4-connectivity).
Rings are rotated to start at their smallest corner in raster order, keep only the vertices where the direction
changes, are closed, and are sorted by that smallest corner.
"""
import numpy as np

_DIRS = [(0, 1), (1, 0), (0, -1), (-1, 0)]   # east, south, west, north as (dy, dx)


def label_rings(labels, start_label=0):
    """-> list of (label, is_hole, [(x, y), ...]) in raster order of each ring's smallest corner (exterior before hole
    at the same corner)."""
    lab = np.asarray(labels)
    H, W = lab.shape

    def at(y, x):
        return lab[y, x] if 0 <= y < H and 0 <= x < W else None

    edges = {}   # label -> {(cy, cx, d)}
    for y in range(H):
        for x in range(W):
            L = int(lab[y, x])
            if L < start_label:
                continue
            e = edges.setdefault(L, set())
            if at(y - 1, x) != L:
                e.add((y, x, 0))
            if at(y, x + 1) != L:
                e.add((y, x + 1, 1))
            if at(y + 1, x) != L:
                e.add((y + 1, x + 1, 2))
            if at(y, x - 1) != L:
                e.add((y + 1, x, 3))
    rings = []
    for L, es in edges.items():
        todo = set(es)
        while todo:
            start = min(todo)                      # smallest corner, east before south at the same corner
            chain = []
            cur = start
            while True:
                todo.discard(cur)
                chain.append(cur)
                cy, cx, d = cur
                ny, nx = cy + _DIRS[d][0], cx + _DIRS[d][1]
                for nd in ((d + 1) % 4, d, (d + 3) % 4):
                    if (ny, nx, nd) in es:
                        cur = (ny, nx, nd)
                        break
                else:  # pragma: no cover
                    raise AssertionError("open ring")
                if cur == start:
                    break
            # the chain may not begin at the ring's smallest corner with the right edge (min over ALL remaining edges of
            # the label can sit on another ring) -- rotate
            k = min(range(len(chain)), key=lambda i: (chain[i][0], chain[i][1]))
            chain = chain[k:] + chain[:k]
            verts = []
            for i, (cy, cx, d) in enumerate(chain):
                if chain[i - 1][2] != d:
                    verts.append((cx, cy))
            verts.append(verts[0])
            area2 = sum(x0 * y1 - x1 * y0 for (x0, y0), (x1, y1) in zip(verts[:-1], verts[1:]))
            # label on the right, y down: an exterior ring runs clockwise on screen = positive shoelace sum in (x, y)
            rings.append(((chain[0][0], chain[0][1], 0 if area2 > 0 else 1), L, area2 < 0, verts))
    rings.sort(key=lambda r: r[0])
    return [(L, hole, verts) for _, L, hole, verts in rings]


def rasterize_rings(rings, H, W, fill=-1):
    """Even-odd fill of (label, is_hole, vertices) back onto an (H, W) raster (vertical edges -> crossing counts)."""
    out = np.full((H, W), fill, np.int64)
    by_label = {}
    for L, hole, verts in rings:
        by_label.setdefault(L, []).append(verts)
    for L, vs in by_label.items():
        diff = np.zeros((H, W + 1), np.int64)
        for verts in vs:
            for (x0, y0), (x1, y1) in zip(verts[:-1], verts[1:]):
                if x0 == x1 and y0 != y1:
                    lo, hi = min(y0, y1), max(y0, y1)
                    diff[lo:hi, x0] += 1
        inside = (np.cumsum(diff, axis=1)[:, :W] % 2) == 1
        out[inside] = L
    return out
