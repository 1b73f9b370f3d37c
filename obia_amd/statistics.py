"""Host-side mirror of the reference's per-segment statistics interface.

Mirrors ``obia.segmentation.segment_statistics.create_objects`` / ``calculate_spectral_stats`` /
``_create_empty_stats_columns`` (segment_statistics.py:392-511, :113-176, :12-110) for the statistics on
the hot path: mean, variance (ddof 0), min, max per band per segment, batched over all segments in one
GPU pass (libobia_hip.so: obia_zonal_stats_f32), plus skewness / kurtosis from a second pass
(obia_zonal_moments_f32).
"""
import ctypes

import numpy as np

from . import _lib

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


def _is_torch(x):
    return torch is not None and isinstance(x, torch.Tensor)


def zonal_stats(raw, labels, bands=None, start_label=1, n_labels=None, ctx=None, moments=False):
    """Per-label statistics of ``raw`` (H,W,C) under the label raster ``labels`` (H,W).

    Returns a dict: ``count`` (N,), ``mean``/``variance`` (N,B) float64, ``min``/``max`` (N,B) float32, with
    N = n_labels (default: max label - start_label + 1) and B = len(bands).  Labels outside
    [start_label, start_label+N) -- e.g. the -1 / 0 of masked pixels -- are ignored; NaN pixels are dropped
    per band; empty segments give NaN (segment_statistics.py:145-162).  NumPy in -> NumPy out; CUDA
    tensors in -> CUDA tensors out.

    ``moments=True`` adds ``skewness`` and ``kurtosis`` (N,B) float64: scipy.stats.skew / kurtosis with their defaults
    (segment_statistics.py:173-175), computed by a second pass with the per-label means as pivots
    (obia_zonal_moments_f32_dev).
    """
    lib = _lib.load()
    if _is_torch(raw):
        if not raw.is_cuda:
            raise ValueError("torch inputs must live on the GPU")
        r = raw.to(torch.float32).contiguous()
        H, W, C = r.shape
        lab = torch.as_tensor(labels, device=r.device).to(torch.int32).contiguous()
        if tuple(lab.shape) != (H, W):
            raise ValueError("labels must have the raster's (H,W) shape")
        if n_labels is None:
            n_labels = int(lab.max().item()) - start_label + 1
        n_labels = max(int(n_labels), 0)
        bl = list(range(C)) if bands is None else [int(b) for b in bands]
        for b in bl:
            if b < 0 or b >= C:
                raise IndexError(f"Band index {b} out of range. Available bands indices: 0 to {C - 1}.")
        B = len(bl)
        barr = np.ascontiguousarray(bl, np.int32)
        dev = r.device
        # (zonal_finalize_kernel writes every entry -- NaN for an empty label -- so the outputs need no fill)
        cnt = torch.empty((n_labels,), dtype=torch.int64, device=dev)
        mean = torch.empty((n_labels, B), dtype=torch.float64, device=dev)
        var = torch.empty_like(mean)
        mn = torch.empty((n_labels, B), dtype=torch.float32, device=dev)
        mx = torch.empty_like(mn)
        c = ctx or _lib.default_context(dev.index or 0)
        torch.cuda.current_stream(dev.index or 0).synchronize()
        _lib.check(lib.obia_zonal_stats_f32_dev(c.handle, r.data_ptr(), lab.data_ptr(), H, W, C, _lib.np_ptr(barr), B,
                                                n_labels, int(start_label), cnt.data_ptr(), mean.data_ptr(),
                                                var.data_ptr(), mn.data_ptr(), mx.data_ptr()))
        out = {"count": cnt, "mean": mean, "variance": var, "min": mn, "max": mx, "bands": bl}
        if moments:
            skew = torch.full_like(mean, float("nan"))
            kurt = torch.full_like(mean, float("nan"))
            _lib.check(lib.obia_zonal_moments_f32_dev(c.handle, r.data_ptr(), lab.data_ptr(), H, W, C, _lib.np_ptr(barr), B,
                                                      n_labels, int(start_label), mean.data_ptr(), skew.data_ptr(),
                                                      kurt.data_ptr()))
            out["skewness"], out["kurtosis"] = skew, kurt
        return out
    r = np.ascontiguousarray(raw, dtype=np.float32)
    if r.ndim != 3:
        raise ValueError("raw must be (H,W,C)")
    H, W, C = r.shape
    lab = np.ascontiguousarray(labels, dtype=np.int32)
    if lab.shape != (H, W):
        raise ValueError("labels must have the raster's (H,W) shape")
    if n_labels is None:
        n_labels = int(lab.max()) - start_label + 1 if lab.size else 0
    n_labels = max(int(n_labels), 0)
    bl = list(range(C)) if bands is None else [int(b) for b in bands]
    for b in bl:
        if b < 0 or b >= C:
            raise IndexError(f"Band index {b} out of range. Available bands indices: 0 to {C - 1}.")
    B = len(bl)
    barr = np.ascontiguousarray(bl, np.int32)
    cnt = np.zeros((n_labels,), np.int64)
    mean = np.full((n_labels, B), np.nan, np.float64)
    var = np.full((n_labels, B), np.nan, np.float64)
    mn = np.full((n_labels, B), np.nan, np.float32)
    mx = np.full((n_labels, B), np.nan, np.float32)
    c = ctx or _lib.default_context(0)
    _lib.check(lib.obia_zonal_stats_f32(c.handle, _lib.np_ptr(r), _lib.np_ptr(lab), H, W, C, _lib.np_ptr(barr), B, n_labels,
                                        int(start_label), _lib.np_ptr(cnt), _lib.np_ptr(mean), _lib.np_ptr(var),
                                        _lib.np_ptr(mn), _lib.np_ptr(mx)))
    out = {"count": cnt, "mean": mean, "variance": var, "min": mn, "max": mx, "bands": bl}
    if moments:
        skew = np.full((n_labels, B), np.nan, np.float64)
        kurt = np.full((n_labels, B), np.nan, np.float64)
        _lib.check(lib.obia_zonal_moments_f32(c.handle, _lib.np_ptr(r), _lib.np_ptr(lab), H, W, C, _lib.np_ptr(barr), B,
                                              n_labels, int(start_label), _lib.np_ptr(skew), _lib.np_ptr(kurt)))
        out["skewness"], out["kurtosis"] = skew, kurt
    return out


TEXTURE_PROPS = ("contrast", "dissimilarity", "homogeneity", "ASM", "energy", "correlation")


def texture_stats(raw, labels, bands=None, start_label=1, n_labels=None, ctx=None):
    """GLCM texture statistics per (label, band): dict prop -> (N, B) float64 for contrast, dissimilarity, homogeneity,
    ASM, energy, correlation (calculate_textural_stats, segment_statistics.py:179-298, on the band plane -- see
    oracle/glcm.py for the one place where this departs from the reference's indexing).  NumPy in -> NumPy out, CUDA
    tensors in -> CUDA tensors out (libobia_hip.so: obia_texture_stats_f32_dev)."""
    if torch is None:
        raise ImportError("obia_amd.statistics.texture_stats needs torch for device memory")
    lib = _lib.load()
    is_t = _is_torch(raw)
    if is_t:
        if not raw.is_cuda:
            raise ValueError("torch inputs must live on the GPU")
        r = raw.to(torch.float32).contiguous()
    else:
        c0 = ctx or _lib.default_context(0)
        r = torch.as_tensor(np.ascontiguousarray(raw, dtype=np.float32), device=f"cuda:{c0.device}")
    if r.dim() != 3:
        raise ValueError("raw must be (H,W,C)")
    H, W, C = r.shape
    lab = torch.as_tensor(labels, device=r.device).to(torch.int32).contiguous()
    if tuple(lab.shape) != (H, W):
        raise ValueError("labels must have the raster's (H,W) shape")
    if n_labels is None:
        n_labels = int(lab.max().item()) - start_label + 1
    n_labels = max(int(n_labels), 0)
    bl = list(range(C)) if bands is None else [int(b) for b in bands]
    for b in bl:
        if b < 0 or b >= C:
            raise IndexError(f"Band index {b} out of range. Available bands indices: 0 to {C - 1}.")
    B = len(bl)
    barr = np.ascontiguousarray(bl, np.int32)
    out = torch.full((6, n_labels, B), float("nan"), dtype=torch.float64, device=r.device)
    c = ctx or _lib.default_context(r.device.index or 0)
    torch.cuda.current_stream(r.device.index or 0).synchronize()
    if n_labels > 0 and B > 0:
        _lib.check(lib.obia_texture_stats_f32_dev(c.handle, r.data_ptr(), lab.data_ptr(), H, W, C, _lib.np_ptr(barr), B,
                                                  n_labels, int(start_label), out.data_ptr()))
    res = {p: (out[i] if is_t else out[i].cpu().numpy()) for i, p in enumerate(TEXTURE_PROPS)}
    res["bands"] = bl
    return res


POINTCLOUD_STATS = ("pai", "fhd", "ch", "mean_intensity", "variance_intensity")


def stats_columns(spectral_bands, textural_bands=(), calc_mean=True, calc_variance=True, calc_min=True, calc_max=True,
                  calc_skewness=True, calc_kurtosis=True, calc_contrast=True, calc_dissimilarity=True,
                  calc_homogeneity=True, calc_ASM=True, calc_energy=True, calc_correlation=True,
                  calc_pai=False, calc_fhd=False, calc_ch=False, calc_mean_intensity=False, calc_variance_intensity=False,
                  geometry=False):
    """Column names and order of the objects table: segment_id, then per spectral band
    mean/variance/min/max/skewness/kurtosis, then per textural band the six GLCM properties, then the point-cloud
    statistics whose flags are set, then ``geometry`` (obia _create_empty_stats_columns, segment_statistics.py:12-110).
    create_objects passes the reference's defaults (all five point-cloud flags True, geometry last)."""
    cols = ["segment_id"]
    spec = [("mean", calc_mean), ("variance", calc_variance), ("min", calc_min), ("max", calc_max),
            ("skewness", calc_skewness), ("kurtosis", calc_kurtosis)]
    for b in spectral_bands:
        cols += [f"b{b}_{name}" for name, on in spec if on]
    tex = [("contrast", calc_contrast), ("dissimilarity", calc_dissimilarity), ("homogeneity", calc_homogeneity),
           ("ASM", calc_ASM), ("energy", calc_energy), ("correlation", calc_correlation)]
    for b in textural_bands:
        cols += [f"b{b}_{name}" for name, on in tex if on]
    cols += [name for name, on in zip(POINTCLOUD_STATS, (calc_pai, calc_fhd, calc_ch, calc_mean_intensity,
                                                         calc_variance_intensity)) if on]
    if geometry:
        cols.append("geometry")
    return cols


def _labels_of(segments):
    """The label raster behind ``segments``: the raster itself, or the table create_segments(as_table=True) returned
    (its ``attrs["labels"]``)."""
    if hasattr(segments, "attrs") and "labels" in getattr(segments, "attrs", {}):
        return segments.attrs["labels"], segments
    return segments, None


def create_objects(segments, image, ept=None, ept_srs=None, spectral_bands=None, textural_bands=None, voxel_resolution=None,
                   calculate_spectral=True, calculate_textural=True, calculate_structural=False, calculate_radiometric=False,
                   calc_mean=True, calc_variance=True, calc_min=True, calc_max=True, calc_skewness=True, calc_kurtosis=True,
                   calc_contrast=True, calc_dissimilarity=True, calc_homogeneity=True, calc_ASM=True, calc_energy=True,
                   calc_correlation=True, calc_pai=True, calc_fhd=True, calc_ch=True, calc_mean_intensity=True,
                   calc_variance_intensity=True, *, start_label=1, geometry=True, ctx=None):
    """Mirror of obia create_objects (segment_statistics.py:392-511): same positional order, same defaults, same column
    set and order -- ``calculate_textural`` defaults to True, the five point-cloud columns are present and NaN (their flags
    default to True while the point-cloud workflow itself raises, :435-439), ``geometry`` comes last.

    ``segments``: the label raster from create_segments (one 4-connected component per label, so "pixels inside polygon p"
    are "pixels carrying label p") or the table create_segments(as_table=True) returned.  ``image``: object
    with ``img_data`` (and optionally ``affine_transformation`` / ``crs``) or the raw (H,W,C) array.  Labels below
    ``start_label`` (the -1 / 0 of masked pixels) get no row; ``segment_id`` = 1..N in ascending label order
    (segment_boundaries.py:76).
    ``geometry``: True -> one polygon per segment from the GPU polygoniser (WKB bytes; shapely geometries in a
    GeoDataFrame when geopandas is installed); False -> the column holds None (skips the polygon pass).
    Spectral statistics are always computed, like the reference's loop (:487-491 does not test ``calculate_spectral``).
    """
    import pandas as pd
    if not (calculate_spectral or calculate_textural or calculate_structural or calculate_radiometric):
        raise ValueError("At least one of 'calculate_spectral', 'calculate_textural', 'calculate_structural', or "
                         "'calculate_radiometric' must be True.")
    if ept is not None or calculate_structural or calculate_radiometric:
        raise NotImplementedError("Point-cloud workflows are temporarily disabled. "
                                  "Use spectral/textural statistics only for now.")
    labels, seg_table = _labels_of(segments)
    img_data = image.img_data if hasattr(image, "img_data") else image
    C = img_data.shape[2]
    if spectral_bands is None:
        spectral_bands = list(range(C))
    if textural_bands is None:
        textural_bands = list(range(C))
    spectral_bands, textural_bands = list(spectral_bands), list(textural_bands)
    st = zonal_stats(img_data, labels, bands=spectral_bands, start_label=start_label, ctx=ctx,
                     moments=bool(calc_skewness or calc_kurtosis))
    if _is_torch(st["count"]):
        st = {k: (v.cpu().numpy() if _is_torch(v) else v) for k, v in st.items()}
    n = st["count"].shape[0]
    present = st["count"] > 0                      # ids of the table: the labels that exist (np.unique order)
    cols = stats_columns(spectral_bands, textural_bands, calc_mean, calc_variance, calc_min, calc_max, calc_skewness,
                         calc_kurtosis, calc_contrast, calc_dissimilarity, calc_homogeneity, calc_ASM, calc_energy,
                         calc_correlation, calc_pai, calc_fhd, calc_ch, calc_mean_intensity, calc_variance_intensity,
                         geometry=True)
    n_rows = int(present.sum())
    data = {"segment_id": np.arange(1, n_rows + 1)}
    for j, b in enumerate(spectral_bands):
        for name, on in (("mean", calc_mean), ("variance", calc_variance), ("min", calc_min), ("max", calc_max),
                         ("skewness", calc_skewness), ("kurtosis", calc_kurtosis)):
            if on:
                data[f"b{b}_{name}"] = st[name][present, j]
    tex_names = [(name, on) for name, on in (("contrast", calc_contrast), ("dissimilarity", calc_dissimilarity),
                                             ("homogeneity", calc_homogeneity), ("ASM", calc_ASM), ("energy", calc_energy),
                                             ("correlation", calc_correlation)) if on]
    if calculate_textural and textural_bands and tex_names:
        tx = texture_stats(img_data, labels, bands=textural_bands, start_label=start_label, n_labels=n, ctx=ctx)
        tx = {k: (v.cpu().numpy() if _is_torch(v) else v) for k, v in tx.items()}
        for j, b in enumerate(textural_bands):
            for name, _ in tex_names:
                data[f"b{b}_{name}"] = tx[name][present, j]
    for c in cols:                                 # textural columns when calculate_textural=False, point-cloud columns: NaN
        if c not in data and c != "geometry":
            data[c] = np.full(n_rows, np.nan)
    geoms = [None] * n_rows
    crs = getattr(image, "crs", None)
    if geometry:
        if seg_table is not None and "geometry" in seg_table and len(seg_table) == n_rows:
            geoms = list(seg_table["geometry"])
        else:
            from .polygons import polygonize
            pt = polygonize(labels, affine_transformation=getattr(image, "affine_transformation", None),
                            start_label=start_label, ctx=ctx)
            if len(pt) != n_rows:
                raise RuntimeError(f"polygoniser returned {len(pt)} polygons for {n_rows} labels")
            geoms = pt.wkb()
    data["geometry"] = geoms
    df = pd.DataFrame(data, columns=cols)
    if geometry:
        try:                                       # the reference returns a GeoDataFrame; without its geo stack: WKB bytes
            import geopandas as gpd
            import shapely
            df = gpd.GeoDataFrame(df.drop(columns="geometry"), geometry=list(shapely.from_wkb(geoms)), crs=crs)[cols]
        except ImportError:
            pass
    df.attrs["crs"] = crs
    return df
