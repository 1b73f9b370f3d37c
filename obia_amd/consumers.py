"""Consumers of the label raster: host-side mirrors of ``slic_edge`` (obia/utils/cost.py:44-48, with its
``normalise``, :21-26) and of ``label_segments`` (obia/utils/utils.py:12-34) that work on the label raster this package
already holds instead of round-tripping through ``segments.gpkg`` (``rasterise_slic_gpkg``, cost.py:51-86, becomes the
identity: the raster of ``segment_id`` IS the label raster)."""
import ctypes
import math

import numpy as np

from . import _lib

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


def _binary_percentile(n, n_ones, q):
    """np.percentile(a, q) (linear interpolation) of an array of n values, n_ones of them 1 and the rest 0."""
    pos = (n - 1) * q / 100.0
    lo = int(math.floor(pos))
    hi = min(lo + 1, n - 1)
    z = n - n_ones
    a_lo, a_hi = (0.0 if lo < z else 1.0), (0.0 if hi < z else 1.0)
    return a_lo + (pos - lo) * (a_hi - a_lo)


def slic_edge(label_img, ctx=None):
    """Edge raster of a label map, float32 in [0, 1] (obia/utils/cost.py:44-48).

    edge = label differs from the pixel below or from the pixel to the right, then ``normalise`` (clip to the 2nd / 98th
    percentile, rescale, NaN -> 0): on a 0/1 image the percentiles follow from the number of edge pixels, which the
    kernel returns with the raster (obia_label_edges_u8_dev).  NumPy in -> NumPy out, CUDA tensor in -> CUDA tensor out.
    """
    if torch is None:
        raise ImportError("obia_amd.consumers needs torch for device memory")
    lib = _lib.load()
    is_t = isinstance(label_img, torch.Tensor)
    if is_t:
        if not label_img.is_cuda:
            raise ValueError("torch inputs must live on the GPU")
        lab = label_img.to(torch.int32).contiguous()
    else:
        c0 = ctx or _lib.default_context(0)
        lab = torch.as_tensor(np.ascontiguousarray(label_img, dtype=np.int32), device=f"cuda:{c0.device}")
    if lab.dim() != 2:
        raise ValueError("label_img must be (H, W)")
    H, W = lab.shape
    c = ctx or _lib.default_context(lab.device.index or 0)
    torch.cuda.current_stream(lab.device.index or 0).synchronize()
    edge = torch.empty((H, W), dtype=torch.uint8, device=lab.device)
    n_edge = ctypes.c_int64(0)
    _lib.check(lib.obia_label_edges_u8_dev(c.handle, lab.data_ptr(), H, W, edge.data_ptr(), ctypes.byref(n_edge)))
    n = H * W
    lo = _binary_percentile(n, int(n_edge.value), 2.0)
    hi = _binary_percentile(n, int(n_edge.value), 98.0)
    e = edge.to(torch.float32)
    if hi == lo:
        out = torch.zeros_like(e)                         # (x - lo) / 0 -> NaN -> 0 (np.nan_to_num)
    else:
        out = (e.clamp(lo, hi) - lo) / (hi - lo)
    return out if is_t else out.cpu().numpy()


def invert_affine(affine_transformation):
    """Inverse of the reference's ``image.affine_transformation`` [a, b, d, e, xoff, yoff] (x' = a x + b y + xoff,
    y' = d x + e y + yoff), in the same layout."""
    a, b, d, e, xoff, yoff = [float(v) for v in affine_transformation]
    det = a * e - b * d
    if det == 0.0:
        raise ValueError("singular affine transformation")
    ia, ib, id_, ie = e / det, -b / det, -d / det, a / det
    return [ia, ib, id_, ie, -(ia * xoff + ib * yoff), -(id_ * xoff + ie * yoff)]


def sample_labels(labels, affine_transformation, points_xy, outside=-1, ctx=None):
    """Label under every point: map coordinates -> pixel (floor of the pixel-corner coordinates) -> label; ``outside``
    for points that fall off the raster.  labels: (H, W) NumPy or CUDA tensor; points_xy: (n, 2) map coordinates."""
    if torch is None:
        raise ImportError("obia_amd.consumers needs torch for device memory")
    lib = _lib.load()
    if isinstance(labels, torch.Tensor):
        lab = labels.to(torch.int32).contiguous()
    else:
        c0 = ctx or _lib.default_context(0)
        lab = torch.as_tensor(np.ascontiguousarray(labels, dtype=np.int32), device=f"cuda:{c0.device}")
    H, W = lab.shape
    pts = np.ascontiguousarray(points_xy, dtype=np.float64).reshape(-1, 2)
    n = pts.shape[0]
    c = ctx or _lib.default_context(lab.device.index or 0)
    d_pts = torch.as_tensor(pts, device=lab.device)
    out = torch.full((max(n, 1),), int(outside), dtype=torch.int32, device=lab.device)
    inv = (ctypes.c_double * 6)(*invert_affine(affine_transformation))
    torch.cuda.current_stream(lab.device.index or 0).synchronize()
    _lib.check(lib.obia_sample_labels_i32_dev(c.handle, lab.data_ptr(), H, W, inv, d_pts.data_ptr(), n, int(outside),
                                              out.data_ptr()))
    return out[:n].cpu().numpy()


def label_segments(labels, affine_transformation, points_xy, classes, start_label=1, ctx=None):
    """Raster version of obia.utils.utils.label_segments (:12-34): a segment whose points all carry one class gets that
    class; a segment that holds points of several classes is reported as mixed; segments without points are dropped.
    Returns ({segment_id: class}, [mixed segment ids]) -- the reference's ``feature_class`` column and ``mixed_segments``
    list.  (A point exactly on a polygon border "intersects" both polygons in the reference; on the raster it belongs
    to the pixel whose corner coordinates floor() to it.)"""
    seg = sample_labels(labels, affine_transformation, points_xy, outside=start_label - 1, ctx=ctx)
    classes = np.asarray(classes)
    labelled, mixed = {}, []
    order = np.argsort(seg, kind="stable")
    seg_s, cls_s = seg[order], classes[order]
    bounds = np.flatnonzero(np.diff(seg_s)) + 1
    for lo, hi in zip(np.concatenate([[0], bounds]), np.concatenate([bounds, [len(seg_s)]])):
        if hi <= lo or seg_s[lo] < start_label:
            continue
        u = np.unique(cls_s[lo:hi])
        if len(u) == 1:
            labelled[int(seg_s[lo])] = u[0].item() if hasattr(u[0], "item") else u[0]
        else:
            mixed.append(int(seg_s[lo]))
    return labelled, mixed
