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
        cnt = torch.zeros((n_labels,), dtype=torch.int64, device=dev)
        mean = torch.full((n_labels, B), float("nan"), dtype=torch.float64, device=dev)
        var = torch.full_like(mean, float("nan"))
        mn = torch.full((n_labels, B), float("nan"), dtype=torch.float32, device=dev)
        mx = torch.full_like(mn, float("nan"))
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


def stats_columns(spectral_bands, textural_bands=(), calc_mean=True, calc_variance=True, calc_min=True, calc_max=True,
                  calc_skewness=True, calc_kurtosis=True, calc_contrast=True, calc_dissimilarity=True,
                  calc_homogeneity=True, calc_ASM=True, calc_energy=True, calc_correlation=True):
    """Column names and order of the objects table: segment_id, then per spectral band
    mean/variance/min/max/skewness/kurtosis, then per textural band the six GLCM properties
    (obia _create_empty_stats_columns, segment_statistics.py:12-110)."""
    cols = ["segment_id"]
    spec = [("mean", calc_mean), ("variance", calc_variance), ("min", calc_min), ("max", calc_max),
            ("skewness", calc_skewness), ("kurtosis", calc_kurtosis)]
    for b in spectral_bands:
        cols += [f"b{b}_{name}" for name, on in spec if on]
    tex = [("contrast", calc_contrast), ("dissimilarity", calc_dissimilarity), ("homogeneity", calc_homogeneity),
           ("ASM", calc_ASM), ("energy", calc_energy), ("correlation", calc_correlation)]
    for b in textural_bands:
        cols += [f"b{b}_{name}" for name, on in tex if on]
    return cols


def create_objects(segments, image, spectral_bands=None, textural_bands=None, calculate_spectral=True,
                   calculate_textural=False, calculate_structural=False, calculate_radiometric=False, ept=None,
                   calc_mean=True, calc_variance=True, calc_min=True, calc_max=True, calc_skewness=True,
                   calc_kurtosis=True, calc_contrast=True, calc_dissimilarity=True, calc_homogeneity=True, calc_ASM=True,
                   calc_energy=True, calc_correlation=True, start_label=1, ctx=None):
    """Array-level mirror of obia create_objects (segment_statistics.py:392-511).

    ``segments``: the label raster from create_segments (one 4-connected component per label, so "pixels
    inside polygon p" are "pixels carrying label p").  ``image``: object with ``img_data`` or
    the raw (H,W,C) array.  Returns a pandas DataFrame whose columns follow the reference's order;
    mean/variance/min/max, skewness/kurtosis and the six GLCM texture statistics come from the GPU passes.
    """
    import pandas as pd
    if not (calculate_spectral or calculate_textural or calculate_structural or calculate_radiometric):
        raise ValueError("At least one of 'calculate_spectral', 'calculate_textural', 'calculate_structural', or "
                         "'calculate_radiometric' must be True.")
    if ept is not None or calculate_structural or calculate_radiometric:
        raise NotImplementedError("Point-cloud workflows are temporarily disabled. "
                                  "Use spectral/textural statistics only for now.")
    img_data = image.img_data if hasattr(image, "img_data") else image
    C = img_data.shape[2]
    if spectral_bands is None:
        spectral_bands = list(range(C))
    tex_bands = list(textural_bands) if (calculate_textural and textural_bands is not None) else (
        list(range(C)) if calculate_textural else [])
    st = zonal_stats(img_data, segments, bands=spectral_bands, start_label=start_label, ctx=ctx,
                     moments=bool(calc_skewness or calc_kurtosis))
    if _is_torch(st["count"]):
        st = {k: (v.cpu().numpy() if _is_torch(v) else v) for k, v in st.items()}
    n = st["count"].shape[0]
    cols = stats_columns(spectral_bands, tex_bands, calc_mean, calc_variance, calc_min, calc_max, calc_skewness,
                         calc_kurtosis, calc_contrast, calc_dissimilarity, calc_homogeneity, calc_ASM, calc_energy,
                         calc_correlation)
    data = {"segment_id": np.arange(1, n + 1)}
    for j, b in enumerate(spectral_bands):
        for name, key, on in (("mean", "mean", calc_mean), ("variance", "variance", calc_variance),
                              ("min", "min", calc_min), ("max", "max", calc_max),
                              ("skewness", "skewness", calc_skewness), ("kurtosis", "kurtosis", calc_kurtosis)):
            if on:
                data[f"b{b}_{name}"] = st[key][:, j]
    if tex_bands:
        tx = texture_stats(img_data, segments, bands=tex_bands, start_label=start_label, n_labels=n, ctx=ctx)
        tx = {k: (v.cpu().numpy() if _is_torch(v) else v) for k, v in tx.items()}
        for j, b in enumerate(tex_bands):
            for name, on in (("contrast", calc_contrast), ("dissimilarity", calc_dissimilarity),
                             ("homogeneity", calc_homogeneity), ("ASM", calc_ASM), ("energy", calc_energy),
                             ("correlation", calc_correlation)):
                if on:
                    data[f"b{b}_{name}"] = tx[name][:, j]
    return pd.DataFrame(data, columns=cols)
