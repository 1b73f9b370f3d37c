"""Label raster -> polygons: host-side mirror of the vectorisation half of ``create_segments``.

The reference (obia/segmentation/segment_boundaries.py:59-77) loops over ``np.unique(segments)``, builds a
full-raster mask per id, runs ``rasterio.features.shapes`` (GDAL polygonize, 4-connected) on it, applies
``image.affine_transformation`` ([a, b, d, e, xoff, yoff], shapely order) and numbers the polygons 1..N in that order.
Here the rings of ALL labels come from one GPU pass (libobia_hip.so: obia_polygon_rings_i32_dev); this module groups
them per label (exterior ring first, then holes), applies the affine transform and hands them over as plain arrays,
GeoJSON-like dicts, WKB, or -- when geopandas / shapely are installed on the user's side -- a GeoDataFrame with the
reference's ``segment_id`` column.
"""
import ctypes
import struct

import numpy as np

from . import _lib

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


class PolygonTable:
    """Rings of every label of a raster.

    ``xy``            (V, 2) float64 vertex coordinates (map coordinates when a transform was given, else pixel-corner
                      coordinates, (0,0) = top-left corner of the raster); vertices only where the outline turns;
                      every ring is closed (first vertex repeated)
    ``ring_offset``   (R + 1,) int64: ring r owns ``xy[ring_offset[r]:ring_offset[r+1]]``
    ``ring_label``    (R,) int32 label of the ring, ``ring_is_hole`` (R,) bool
    Rings are ordered by label, and within a label the exterior ring comes first, then its holes (each label of a
    connectivity-enforced label map is one 4-connected component, hence one exterior ring).
    ``labels``        (N,) the distinct labels in ascending order -- polygon i gets ``segment_id = i + 1`` like the
                      reference's ``range(1, len(gdf) + 1)``; ``poly_ring_start`` (N + 1,) indexes the rings of polygon i.
    """

    def __init__(self, xy, ring_offset, ring_label, ring_is_hole, ring_part=None):
        self.xy, self.ring_offset, self.ring_label, self.ring_is_hole = xy, ring_offset, ring_label, ring_is_hole
        self.labels, first = np.unique(ring_label, return_index=True)
        self.poly_ring_start = np.append(first, len(ring_label)).astype(np.int64)
        # ring_part[r]: index of the exterior ring whose polygon ring r belongs to (itself for exterior rings)
        self.ring_part = ring_part if ring_part is not None else first[np.searchsorted(self.labels, ring_label)]

    def __len__(self):
        return len(self.labels)

    def rings_of(self, i):
        """[(is_hole, (n, 2) coordinates), ...] of polygon i (exterior first)."""
        out = []
        for r in range(self.poly_ring_start[i], self.poly_ring_start[i + 1]):
            out.append((bool(self.ring_is_hole[r]), self.xy[self.ring_offset[r]:self.ring_offset[r + 1]]))
        return out

    def _parts(self, i):
        """Polygons of label i as lists of rings: one part per exterior ring (a label that is not connected has several)."""
        lo, hi = self.poly_ring_start[i], self.poly_ring_start[i + 1]
        parts = {}
        for r in range(lo, hi):
            ring = self.xy[self.ring_offset[r]:self.ring_offset[r + 1]]
            parts.setdefault(int(self.ring_part[r]), []).append((bool(self.ring_is_hole[r]), ring))
        return [[ring for hole, ring in sorted(v, key=lambda t: t[0])] for k, v in sorted(parts.items())]

    def geojson_features(self):
        """GeoJSON-like Feature dicts with ``segment_id`` = 1..N in label order (segment_boundaries.py:76)."""
        feats = []
        for i in range(len(self)):
            parts = self._parts(i)
            if len(parts) == 1:
                geom = {"type": "Polygon", "coordinates": [r.tolist() for r in parts[0]]}
            else:
                geom = {"type": "MultiPolygon", "coordinates": [[r.tolist() for r in part] for part in parts]}
            feats.append({"type": "Feature", "properties": {"segment_id": i + 1, "label": int(self.labels[i])},
                          "geometry": geom})
        return feats

    def wkb(self):
        """Little-endian WKB of every polygon (list of bytes), ready for ``shapely.from_wkb`` / a GeoPackage writer."""
        out = []
        for i in range(len(self)):
            parts = self._parts(i)

            def poly_bytes(rings):
                b = struct.pack("<BII", 1, 3, len(rings))
                for r in rings:
                    b += struct.pack("<I", len(r)) + np.ascontiguousarray(r, "<f8").tobytes()
                return b
            if len(parts) == 1:
                out.append(poly_bytes(parts[0]))
            else:
                out.append(struct.pack("<BII", 1, 6, len(parts)) + b"".join(poly_bytes(p) for p in parts))
        return out

    def to_geodataframe(self, crs=None):
        """GeoDataFrame(geometry, segment_id) as create_segments returns it; needs geopandas + shapely (not part of this
        package's requirements -- the reference's own stack provides them)."""
        import geopandas as gpd
        import shapely
        gdf = gpd.GeoDataFrame(geometry=list(shapely.from_wkb(self.wkb())), crs=crs)
        gdf["segment_id"] = range(1, len(gdf) + 1)
        return gdf

    def areas(self):
        """Signed shoelace area per ring in the units of ``xy`` (exterior and holes have opposite signs)."""
        x, y = self.xy[:, 0], self.xy[:, 1]
        cross = x[:-1] * y[1:] - x[1:] * y[:-1]
        cs = np.concatenate([[0.0], np.cumsum(cross)])
        lo, hi = self.ring_offset[:-1], self.ring_offset[1:] - 1      # the closing edge is part of the ring
        return 0.5 * (cs[hi] - cs[lo])


def _point_in_ring(p, ring):
    x, y = ring[:, 0], ring[:, 1]
    x0, y0, x1, y1 = x[:-1], y[:-1], x[1:], y[1:]
    cond = (y0 > p[1]) != (y1 > p[1])
    with np.errstate(divide="ignore", invalid="ignore"):
        xi = x0 + (p[1] - y0) * (x1 - x0) / (y1 - y0)
    return bool(np.count_nonzero(cond & (p[0] < xi)) & 1)


def polygonize(labels, affine_transformation=None, start_label=0, ctx=None):
    """Polygons of a label raster (``create_segments`` back half, segment_boundaries.py:59-77).

    ``labels``: (H, W) integer raster, NumPy or CUDA tensor; pixels with a label below ``start_label`` (the reference
    writes -1 where the mask is 0 and skips that id) get no polygon.  ``affine_transformation``: the reference's
    ``image.affine_transformation`` = [a, b, d, e, xoff, yoff] (x' = a*x + b*y + xoff, y' = d*x + e*y + yoff applied to
    pixel-corner coordinates); None keeps pixel-corner coordinates.  Returns a :class:`PolygonTable`.
    """
    if torch is None:
        raise ImportError("obia_amd.polygons needs torch for device memory")
    lib = _lib.load()
    if isinstance(labels, torch.Tensor):
        if not labels.is_cuda:
            raise ValueError("torch inputs must live on the GPU")
        lab = labels.to(torch.int32).contiguous()
    else:
        c0 = ctx or _lib.default_context(0)
        lab = torch.as_tensor(np.ascontiguousarray(labels, dtype=np.int32), device=f"cuda:{c0.device}")
    if lab.dim() != 2:
        raise ValueError("labels must be (H, W)")
    H, W = lab.shape
    dev = lab.device
    c = ctx or _lib.default_context(dev.index or 0)
    torch.cuda.current_stream(dev.index or 0).synchronize()
    n_r, n_v = ctypes.c_int64(0), ctypes.c_int64(0)
    _lib.check(lib.obia_polygon_count_i32_dev(c.handle, lab.data_ptr(), H, W, int(start_label), ctypes.byref(n_r),
                                              ctypes.byref(n_v)))
    R, V = int(n_r.value), int(n_v.value)
    ring_label = torch.empty((max(R, 1),), dtype=torch.int32, device=dev)
    ring_hole = torch.empty((max(R, 1),), dtype=torch.uint8, device=dev)
    ring_off = torch.zeros((R + 1,), dtype=torch.int64, device=dev)
    xy = torch.empty((max(V, 1), 2), dtype=torch.int32, device=dev)
    _lib.check(lib.obia_polygon_rings_i32_dev(c.handle, lab.data_ptr(), H, W, int(start_label), R, V, ring_label.data_ptr(),
                                              ring_hole.data_ptr(), ring_off.data_ptr(), xy.data_ptr(), ctypes.byref(n_r),
                                              ctypes.byref(n_v)))
    xyf = xy[:V].to(torch.float64)
    if affine_transformation is not None:
        a, b, d, e, xoff, yoff = [float(v) for v in affine_transformation]
        xyf = torch.stack([a * xyf[:, 0] + b * xyf[:, 1] + xoff, d * xyf[:, 0] + e * xyf[:, 1] + yoff], dim=1)
    xy_h = xyf.cpu().numpy()
    rl = ring_label[:R].cpu().numpy()
    rh = ring_hole[:R].cpu().numpy().astype(bool)
    ro = ring_off.cpu().numpy()
    # group: by label, exterior before holes, otherwise raster order of the rings' smallest corners (stable sort)
    order = np.lexsort((rh, rl)) if R else np.zeros((0,), np.int64)
    lens = (ro[1:] - ro[:-1])[order]
    new_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    idx = np.concatenate([np.arange(ro[r], ro[r + 1]) for r in order]) if R and R < 64 else None
    if R:
        if idx is None:      # vectorised gather of the vertex ranges
            starts = ro[:-1][order]
            idx = np.repeat(starts - new_off[:-1], lens) + np.arange(new_off[-1])
        xy_h = xy_h[idx]
    rl, rh = rl[order], rh[order]
    ring_part = _assign_parts(xy[:V].cpu().numpy()[idx] if R else np.zeros((0, 2), np.int32), new_off, rl, rh)
    return PolygonTable(xy_h, new_off, rl, rh, ring_part)


def _assign_parts(xy_pix, off, ring_label, ring_hole):
    """Exterior ring of every ring, decided on pixel-corner coordinates.  A connectivity-enforced label map has one
    exterior ring per label (every hole belongs to it); a label with several 4-connected parts gets its holes assigned
    to the smallest exterior ring that contains them."""
    R = len(ring_label)
    part = np.arange(R, dtype=np.int64)
    if R == 0:
        return part
    labels, first, counts = np.unique(ring_label, return_index=True, return_counts=True)
    n_ext = np.add.reduceat((~ring_hole).astype(np.int64), first)
    single = np.repeat(n_ext == 1, counts)
    part[single] = np.repeat(first, counts)[single]          # the exterior ring is the first ring of its label
    for li in np.nonzero(n_ext > 1)[0]:
        lo, hi = first[li], first[li] + counts[li]
        exts = [r for r in range(lo, hi) if not ring_hole[r]]
        rings = {r: xy_pix[off[r]:off[r + 1]].astype(np.float64) for r in range(lo, hi)}
        area = {r: abs(_shoelace(rings[r])) for r in exts}
        for r in range(lo, hi):
            if not ring_hole[r]:
                continue
            a, b = rings[r][0], rings[r][1]
            d = b - a
            n = np.array([d[1], -d[0]]) / max(np.hypot(*d), 1e-30)
            p = 0.5 * (a + b) + 0.25 * n                     # just beside the middle of the first edge: off the pixel grid
            inside = [e for e in exts if _point_in_ring(p, rings[e])]
            part[r] = min(inside, key=lambda e: area[e]) if inside else exts[0]
    return part


def _shoelace(ring):
    x, y = ring[:, 0], ring[:, 1]
    return 0.5 * float(np.sum(x[:-1] * y[1:] - x[1:] * y[:-1]))
