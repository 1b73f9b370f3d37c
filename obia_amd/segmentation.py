"""Host-side mirror of the reference's segmentation interface for the hot path.

Mirrors (same names, argument meaning and error behaviour):
  * ``skimage.segmentation.slic`` as obia calls it (obia/segmentation/segment_boundaries.py:48-51)
  * ``obia.segmentation.segment_boundaries.normalize_band / create_segments`` (:11-16, :18-78)
  * ``obia.segmentation.segment.segment / Segments``                          (segment.py:10-93)
All arithmetic runs in libobia_hip.so (hand-written HIP, gfx950); this module only checks arguments,
moves arrays and shapes the results.  There is no CPU path.
"""
import ctypes
import warnings

import numpy as np

from . import _lib

try:  # torch is plumbing only: device tensors in, device tensors out
    import torch
except Exception:  # pragma: no cover
    torch = None

_SLIC_KWARGS = ("n_segments", "compactness", "max_num_iter", "max_iter", "sigma", "spacing", "convert2lab",
                "enforce_connectivity", "min_size_factor", "max_size_factor", "slic_zero", "start_label", "mask",
                "channel_axis", "multichannel", "exit_on_fixed_point")


def _is_torch(x):
    return torch is not None and isinstance(x, torch.Tensor)


def make_params(n_segments=100, compactness=10.0, max_num_iter=10, convert2lab=None, enforce_connectivity=True,
                min_size_factor=0.5, max_size_factor=3, slic_zero=False, start_label=1, normalize_bands=False,
                exit_on_fixed_point=False, sigma=0, spacing=None):
    p = _lib.SlicParams()
    p.n_segments = int(n_segments)
    p.compactness = float(compactness)
    p.max_num_iter = int(max_num_iter)
    p.convert2lab = -1 if convert2lab is None else int(bool(convert2lab))
    p.enforce_connectivity = int(bool(enforce_connectivity))
    p.min_size_factor = float(min_size_factor)
    p.max_size_factor = float(max_size_factor)
    p.slic_zero = int(bool(slic_zero))
    p.start_label = int(start_label)
    p.normalize_bands = int(bool(normalize_bands))
    p.exit_on_fixed_point = int(bool(exit_on_fixed_point))
    p.reserved = 0
    for i, v in enumerate(sigma_zyx(sigma, spacing)):
        p.sigma_zyx[i] = v
    for i, v in enumerate(spacing_zyx(spacing)):
        p.spacing_zyx[i] = v
    return p


def spacing_zyx(spacing):
    """scikit-image's `spacing` (voxel size per axis: depth, row, column -- or (row, column), the form scikit-image >= 0.19
    takes for a 2-D image, which the reference's pin `scikit-image>=0.23.2` implies) in the image's dtype, float32."""
    if spacing is None:
        return [1.0, 1.0, 1.0]
    s = np.ascontiguousarray(spacing, dtype=np.float32).ravel()
    if s.shape == (2,):   # scikit-image >= 0.19 on a 2-D image: (row, column), the one-plane depth axis gets spacing 1
        s = np.insert(s, 0, np.float32(1.0))
    if s.shape != (3,):
        raise ValueError("spacing: a (row, column) or (depth, row, column) sequence")
    if not np.all(np.isfinite(s)) or np.any(s <= 0):
        raise ValueError("spacing must be positive and finite")
    return [float(v) for v in s]


def sigma_zyx(sigma, spacing=None):
    """scikit-image's reading of `sigma` (slic_superpixels.py): the widths live in the image's dtype (float32); a number is the width on
    every axis of the (1, H, W, C) image -- the one-plane depth axis included -- DIVIDED by the spacing, a sequence is taken as
    (depth, row, column) as it is."""
    if sigma is None:
        return [0.0, 0.0, 0.0]
    if np.isscalar(sigma):
        s = np.array([sigma, sigma, sigma], dtype=np.float32)
        s /= np.asarray(spacing_zyx(spacing), np.float32)
    else:
        s = np.array(sigma, dtype=np.float32).ravel()
        if s.shape == (2,):   # scikit-image >= 0.19 on a 2-D image: (row, column), no smoothing along the one-plane depth axis
            s = np.insert(s, 0, np.float32(0.0))
        if s.shape != (3,):
            raise ValueError("sigma: a number, a (row, column) or a (depth, row, column) sequence")
    if not np.all(s >= 0):
        raise ValueError("sigma must be >= 0")
    return [float(v) for v in s]


def _check_common(sigma, spacing, channel_axis, multichannel, sigma_ok=False):
    if not sigma_ok and np.any(np.asarray(sigma) != 0):
        raise NotImplementedError("sigma != 0 (Gaussian pre-smoothing) is not implemented for this operator; obia never sets it")
    if not sigma_ok and spacing is not None:
        raise NotImplementedError("spacing is not implemented for this operator")
    if channel_axis not in (-1, None, 2):
        raise NotImplementedError("channel_axis must be -1 (band-interleaved (H,W,C) rasters)")
    if multichannel is not None and not multichannel:
        raise NotImplementedError("multichannel=False (3-D volumes) is not implemented")


def slic(image, n_segments=100, compactness=10.0, max_num_iter=10, sigma=0, spacing=None, convert2lab=None,
         enforce_connectivity=True, min_size_factor=0.5, max_size_factor=3, slic_zero=False, start_label=1,
         mask=None, *, channel_axis=-1, max_iter=None, multichannel=None, exit_on_fixed_point=False, ctx=None,
         seeds=None, _normalize_bands=False, _stage="full"):
    """Drop-in for ``skimage.segmentation.slic`` on 2-D multichannel rasters (the call at
    obia/segmentation/segment_boundaries.py:51), executed on the GPU.

    image : (H,W) or (H,W,C) array (NumPy: host call, returns ``np.int64`` labels like scikit-image's
        ``intp``; torch CUDA tensor: device call, returns an ``int32`` CUDA tensor).  Any strides.
        Computation is float32, obia's raster dtype (obia/handlers/geotif.py:100).
    mask : (H,W) bool/uint8, optional.  Keeps the reference's maskSLIC structure (spatial-only pre-pass)
        with the deterministic masked-grid seeding of DESIGN.md.
    sigma : number or (depth, row, column) sequence -- scikit-image's Gaussian pre-smoothing (``ndi.gaussian_filter`` on the
        (1, H, W, C) image after the Lab conversion, before ``* 1/compactness``; a number is the same width on all three axes,
        divided by ``spacing``).  Pinned bit for bit on SciPy's filter (tests/golden/sigma*.npz).
    spacing : (depth, row, column) sequence -- voxel size per axis as in scikit-image: the row / column differences of the distance
        term are scaled by it.  Anything but (1, 1, 1) takes the sweep's direct path (exact, an order of magnitude slower).
    seeds : ``(centroids_yx (K,2), steps)`` -- initial centroids to use instead of the library's seeding rule, e.g.
        the output of scikit-image's own ``_get_mask_centroids`` / ``_get_grid_centroids`` (``steps`` as returned
        there: 3 values, depth axis first, or 2 values (y, x)).  CUDA tensor images only.  Not a scikit-image argument.
    exit_on_fixed_point : stop sweeping once a sweep starts from centroids bit-identical to the previous sweep's
        (every later sweep would reproduce the same labels): same result as all ``max_num_iter`` sweeps, less
        work on rasters that converge early (e.g. compactness 10 on [0,1] features).  Not a scikit-image argument.
    Raises ValueError / NotImplementedError like the reference for bad / unsupported arguments.
    """
    _check_common(sigma, spacing, channel_axis, multichannel, sigma_ok=True)
    if max_iter is not None:            # scikit-image < 0.19 keyword
        max_num_iter = max_iter
    if start_label not in (0, 1):
        raise ValueError("start_label should be 0 or 1.")
    params = make_params(n_segments, compactness, max_num_iter, convert2lab, enforce_connectivity, min_size_factor,
                         max_size_factor, slic_zero, start_label, _normalize_bands, exit_on_fixed_point, sigma, spacing)
    lib = _lib.load()
    n_out = ctypes.c_int(0)
    if _is_torch(image):
        if not image.is_cuda:
            raise ValueError("torch inputs must live on the GPU; pass a NumPy array for host data")
        img = image if image.dim() == 3 else image[..., None]
        img = img.to(torch.float32).contiguous()
        H, W, C = img.shape
        m = None
        if mask is not None:
            m = torch.as_tensor(mask, device=img.device)
            if tuple(m.shape) != (H, W):
                raise ValueError("image and mask should have the same shape.")
            m = _lib.mask_bytes(m)
        dev = img.device.index or 0
        c = ctx or _lib.default_context(dev)
        torch.cuda.current_stream(dev).synchronize()
        out = torch.empty((H, W), dtype=torch.int32, device=img.device)
        if seeds is not None:
            yx = np.ascontiguousarray(seeds[0], np.float64)
            if yx.ndim != 2 or yx.shape[1] != 2:
                raise ValueError("seeds[0] must be (K, 2) centroid positions (y, x)")
            st = [float(v) for v in np.ravel(seeds[1])]
            if len(st) == 2:
                st = [1.0] + st
            if len(st) != 3:
                raise ValueError("seeds[1] must hold 2 (y, x) or 3 (z, y, x) steps")
            sd = _lib.SlicSeeds()
            sd.yx, sd.n = yx.ctypes.data, yx.shape[0]
            sd.steps_zyx[:] = st
            _lib.check(lib.obia_slic_seeded_f32_dev(c.handle, img.data_ptr(), H, W, C, m.data_ptr() if m is not None else None,
                                                    ctypes.byref(params), ctypes.byref(sd), {"full": 0, "pre": 1}[_stage],
                                                    out.data_ptr(), ctypes.byref(n_out)))
            return out
        fn = {"full": lib.obia_slic_f32_dev, "pre": lib.obia_slic_assign_only_f32_dev}[_stage]
        _lib.check(fn(c.handle, img.data_ptr(), H, W, C, m.data_ptr() if m is not None else None,
                      ctypes.byref(params), out.data_ptr(), ctypes.byref(n_out)))
        return out
    img = np.asarray(image)
    if img.ndim == 2:
        img = img[..., None]
    if img.ndim != 3:
        raise ValueError("image must be (H,W) or (H,W,C)")
    img = np.ascontiguousarray(img, dtype=np.float32)
    H, W, C = img.shape
    m = None
    if mask is not None:
        m = np.asarray(mask)
        if m.shape != (H, W):
            raise ValueError("image and mask should have the same shape.")
        m = np.ascontiguousarray(m != 0, dtype=np.uint8)
    c = ctx or _lib.default_context(0)
    out = np.empty((H, W), np.int32)
    if _stage != "full" or seeds is not None:
        raise ValueError("stage-level and seeded calls need device tensors")
    _lib.check(lib.obia_slic_f32(c.handle, _lib.np_ptr(img), H, W, C, _lib.np_ptr(m), ctypes.byref(params),
                                 _lib.np_ptr(out), ctypes.byref(n_out)))
    return out.astype(np.int64)


def quickshift(image, ratio=1.0, kernel_size=5, max_dist=10, return_tree=False, sigma=0, convert2lab=True,
               rng=42, *, random_seed=None, channel_axis=-1, ctx=None, _normalize_bands=False):
    """Drop-in for ``skimage.segmentation.quickshift`` as obia calls it
    (obia/segmentation/segment_boundaries.py:48-49), on the GPU, float64 arithmetic.

    The densities get the tie-breaking noise scikit-image adds: ``rng`` (scikit-image >= 0.21:
    ``np.random.default_rng(rng).normal(scale=1e-5)``) or, when ``random_seed`` is given, the legacy
    ``np.random.RandomState(random_seed)`` stream of scikit-image < 0.21 (the stream of the golden vectors).
    Drawing 6.7e7 normals on the host is 3/4 of the wall time at 8192 x 8192; ``rng="device"`` (CUDA tensors only)
    draws the noise on the GPU instead (torch's Philox stream seeded with 42) -- NOT NumPy's stream, so results can differ
    from scikit-image's where two densities tie to within 1e-5 (flat regions), nowhere else.
    NumPy array in -> ``np.int64`` labels out; CUDA tensor in -> int32 CUDA tensor out.
    """
    if return_tree:
        raise NotImplementedError("return_tree=True is not implemented")
    if not np.isscalar(sigma) or not (float(sigma) >= 0.0):
        raise ValueError("sigma: a number >= 0 (the width of the Gaussian pre-smoothing on both raster axes)")
    if channel_axis not in (-1, None, 2):
        raise NotImplementedError("channel_axis must be -1")
    if kernel_size < 1:
        raise ValueError("`kernel_size` should be >= 1.")
    # any band count up to 16 and any kernel_size: 1 / 3 / 4 bands with kernel_size <= 5 take the LDS-staged kernel, other
    # calls the same arithmetic on global memory (csrc/quickshift.hip)
    lib = _lib.load()
    n_out = ctypes.c_int(0)
    is_t = _is_torch(image)
    shape = tuple(image.shape)
    H, W = shape[0], shape[1]
    C = 1 if len(shape) == 2 else shape[2]
    if convert2lab and C != 3:
        raise ValueError("Only RGB images can be converted to Lab space.")
    device_noise = isinstance(rng, str) and rng == "device"
    if device_noise:
        if not is_t:
            raise ValueError('rng="device" needs a CUDA tensor input')
        noise = None
    elif random_seed is not None:
        noise = np.random.RandomState(random_seed).normal(scale=0.00001, size=(H, W))
    else:
        noise = (rng if isinstance(rng, np.random.Generator) else np.random.default_rng(rng)).normal(scale=0.00001, size=(H, W))
    if noise is not None:
        noise = np.ascontiguousarray(noise, np.float64)
    if is_t:
        if not image.is_cuda:
            raise ValueError("torch inputs must live on the GPU; pass a NumPy array for host data")
        img = (image if image.dim() == 3 else image[..., None]).to(torch.float32).contiguous()
        dev = img.device.index or 0
        c = ctx or _lib.default_context(dev)
        if device_noise:
            g = torch.Generator(device=img.device).manual_seed(42)
            nz = torch.randn((H, W), dtype=torch.float64, device=img.device, generator=g) * 0.00001
        else:
            nz = torch.as_tensor(noise, device=img.device)
        out = torch.empty((H, W), dtype=torch.int32, device=img.device)
        torch.cuda.current_stream(dev).synchronize()
        _lib.check(lib.obia_quickshift_f32_dev(c.handle, img.data_ptr(), H, W, C, float(ratio), float(kernel_size),
                                               float(max_dist), float(sigma), int(bool(convert2lab)), nz.data_ptr(),
                                               int(bool(_normalize_bands)), out.data_ptr(), ctypes.byref(n_out)))
        return out
    img = np.asarray(image)
    if img.ndim == 2:
        img = img[..., None]
    img = np.ascontiguousarray(img, dtype=np.float32)
    c = ctx or _lib.default_context(0)
    out = np.empty((H, W), np.int32)
    _lib.check(lib.obia_quickshift_f32(c.handle, _lib.np_ptr(img), H, W, C, float(ratio), float(kernel_size), float(max_dist),
                                       float(sigma), int(bool(convert2lab)), _lib.np_ptr(noise), int(bool(_normalize_bands)),
                                       _lib.np_ptr(out), ctypes.byref(n_out)))
    return out.astype(np.int64)


def enforce_connectivity(labels, min_size, max_size, start_label=1, ctx=None):
    """Connectivity enforcement alone on an int32 CUDA label tensor (stage-level parity hook)."""
    if not _is_torch(labels) or not labels.is_cuda:
        raise ValueError("enforce_connectivity needs an int32 CUDA tensor")
    lab = labels.to(torch.int32).contiguous()
    H, W = lab.shape
    dev = lab.device.index or 0
    c = ctx or _lib.default_context(dev)
    torch.cuda.current_stream(dev).synchronize()
    out = torch.empty_like(lab)
    n = ctypes.c_int(0)
    _lib.check(_lib.load().obia_enforce_connectivity_i32_dev(c.handle, lab.data_ptr(), H, W, int(min_size), int(max_size),
                                                              int(start_label), out.data_ptr(), ctypes.byref(n)))
    return out, n.value


def normalize_band(band):
    """obia.segmentation.segment_boundaries.normalize_band (segment_boundaries.py:11-16), host helper kept
    for callers that use it directly; the segmentation path normalises on the GPU."""
    band = np.asarray(band)
    return (band - np.min(band)) / (np.max(band) - np.min(band))


def segments_table(labels, image=None, start_label=0, ctx=None):
    """The table the reference's create_segments returns (segment_boundaries.py:59-77): one row per segment with
    ``geometry`` (polygon in map coordinates through ``image.affine_transformation``; WKB bytes, or shapely geometries in a
    GeoDataFrame when geopandas is installed) and ``segment_id`` = 1..N in ascending label order; labels below
    ``start_label`` (the -1 of masked pixels) get no row.  The label raster rides along in ``attrs["labels"]`` so that
    create_objects can take the table like the reference takes its GeoDataFrame."""
    import pandas as pd
    from .polygons import polygonize
    pt = polygonize(labels, affine_transformation=getattr(image, "affine_transformation", None), start_label=start_label, ctx=ctx)
    wkb = pt.wkb()
    crs = getattr(image, "crs", None)
    try:
        import geopandas as gpd
        import shapely
        df = gpd.GeoDataFrame(geometry=list(shapely.from_wkb(wkb)), crs=crs)
    except ImportError:
        df = pd.DataFrame({"geometry": wkb})
    df["segment_id"] = range(1, len(df) + 1)
    df.attrs["labels"] = labels
    df.attrs["crs"] = crs
    return df


def create_segments(image, segmentation_bands=None, method="slic", inplace_normalize=False, as_table=False, ctx=None, **kwargs):
    """Mirror of obia create_segments (segment_boundaries.py:18-78).

    ``image``: object with ``img_data`` (H,W,C) float32 (obia ``Image``), or the array itself.
    Returns the label raster (H,W) int64: labels consecutive from ``start_label`` (default 1), and -1
    where ``mask == 0`` (segment_boundaries.py:55-57).  ``as_table=True`` returns what the reference returns instead --
    the ``geometry`` / ``segment_id`` table of :func:`segments_table` (segment_boundaries.py:59-77), polygons from the
    GPU polygoniser.

    Every band of the raster is min-max normalised before band selection, as the reference does
    (:32-33) -- on the GPU, on a private copy: the caller's ``img_data`` is NOT mutated unless
    ``inplace_normalize=True`` reproduces that side effect on the host.
    """
    img_data = image.img_data if hasattr(image, "img_data") else image
    num_bands = img_data.shape[2]
    if segmentation_bands is None:
        segmentation_bands = list(range(num_bands))
    for band in segmentation_bands:
        if band >= num_bands or band < 0:
            raise IndexError(f"Band index {band} out of range. Available bands indices: 0 to {num_bands - 1}.")
    if method not in ("slic", "quickshift"):
        raise Exception("An unknown segmentation method was requested.")
    if _is_torch(img_data):
        sel = img_data[:, :, list(segmentation_bands)]
    else:
        sel = np.asarray(img_data)[:, :, list(segmentation_bands)]
    if method == "quickshift":
        qs_kw = ("ratio", "kernel_size", "max_dist", "return_tree", "sigma", "convert2lab", "rng", "random_seed", "channel_axis")
        unknown = [k for k in kwargs if k not in qs_kw]
        if unknown:
            raise TypeError(f"quickshift() got an unexpected keyword argument '{unknown[0]}'")
        segments = quickshift(sel, ctx=ctx, _normalize_bands=True, **kwargs)
        if inplace_normalize and not _is_torch(img_data):
            for i in range(num_bands):
                img_data[:, :, i] = normalize_band(img_data[:, :, i])
        return segments_table(segments, image, start_label=0, ctx=ctx) if as_table else segments
    unknown = [k for k in kwargs if k not in _SLIC_KWARGS]
    if unknown:
        raise TypeError(f"slic() got an unexpected keyword argument '{unknown[0]}'")
    kwargs.setdefault("start_label", 1)   # scikit-image >= 0.19 default (pyproject.toml:23 pins >= 0.23.2)
    # normalisation is per band, so selecting first and normalising the selected bands is identical
    segments = slic(sel, ctx=ctx, _normalize_bands=True, **kwargs)
    if inplace_normalize and not _is_torch(img_data):
        for i in range(num_bands):
            img_data[:, :, i] = normalize_band(img_data[:, :, i])
    mask = kwargs.get("mask", None)
    if mask is not None:
        if _is_torch(segments):
            segments[torch.as_tensor(mask, device=segments.device) == 0] = -1
        else:
            segments[np.asarray(mask) == 0] = -1
    # the reference skips only the id -1 (:62-64); with start_label=1 and no mask every label >= 1, with start_label=0 the
    # label 0 is a segment like any other
    return segments_table(segments, image, start_label=0 if mask is None else kwargs["start_label"], ctx=ctx) if as_table else segments


class Segments:
    """Result holder mirroring obia.segmentation.segment.Segments (segment.py:10-60): ``_segments`` is the
    label raster, ``segments`` the per-segment objects table."""

    def __init__(self, _segments, segments, method, **kwargs):
        self._segments = _segments
        self.segments = segments
        self.method = method
        self.params = dict(kwargs)

    def polygons(self, affine_transformation=None, start_label=1, ctx=None):
        """Polygons of the label raster (the GeoDataFrame geometry of the reference's create_segments,
        segment_boundaries.py:59-77): obia_amd.polygons.polygonize on ``_segments``.  Pass the image's
        ``affine_transformation`` ([a, b, d, e, xoff, yoff]) for map coordinates."""
        from .polygons import polygonize
        return polygonize(self._segments, affine_transformation=affine_transformation, start_label=start_label, ctx=ctx)

    def write_segments(self, file_path):
        """segment.py:55-60 (``self.segments.to_file(file_path)``): GeoDataFrame.to_file when geopandas is there; else a
        GeoPackage written by obia_amd.geopackage (``.gpkg``) or a CSV without the geometry."""
        if hasattr(self.segments, "to_file"):
            self.segments.to_file(file_path)
        elif str(file_path).lower().endswith(".gpkg") and "geometry" in self.segments and len(self.segments) \
                and isinstance(self.segments["geometry"].iloc[0], (bytes, bytearray)):
            from .geopackage import write_geopackage
            cols = {c: self.segments[c].to_numpy() for c in self.segments.columns if c != "geometry"}
            write_geopackage(file_path, list(self.segments["geometry"]), cols, table="segments",
                             srs_epsg=_epsg_of(self.segments.attrs.get("crs")))
        else:
            self.segments.drop(columns=[c for c in ("geometry",) if c in self.segments]).to_csv(file_path, index=False)


def _epsg_of(crs):
    """EPSG code of ``image.crs`` when it is spelled as one ("EPSG:32610", 32610); None otherwise (no pyproj here)."""
    if crs is None:
        return None
    if isinstance(crs, int):
        return crs
    txt = str(crs).strip().upper()
    if txt.startswith("EPSG:") and txt[5:].isdigit():
        return int(txt[5:])
    return int(txt) if txt.isdigit() else None


def segment(image, segmentation_bands=None, statistics_bands=None, method="slic", calc_mean=True, calc_variance=True,
            calc_skewness=True, calc_kurtosis=True, calc_contrast=True, calc_dissimilarity=True, calc_homogeneity=True,
            calc_ASM=True, calc_energy=True, calc_correlation=True, ctx=None, **kwargs):
    """Mirror of obia.segmentation.segment.segment (segment.py:63-93): create_segments then create_objects.

    Statistics are taken from the RAW raster values (the reference re-reads them from the file,
    utils/utils.py:47; here ``image.img_data`` is still raw because create_segments does not mutate it).
    """
    from .statistics import create_objects
    labels = create_segments(image, segmentation_bands=segmentation_bands, method=method, ctx=ctx, **kwargs)
    # ids of the objects table: every label the segmentation produced.  quickshift numbers from 0; slic from start_label
    # (0 or 1), and with a mask the masked pixels carry -1 (segment_boundaries.py:55-64 skips only that id)
    first = 0 if method == "quickshift" else int(kwargs.get("start_label", 1))
    objects = create_objects(labels, image, spectral_bands=statistics_bands, calc_mean=calc_mean,
                             calc_variance=calc_variance, calc_skewness=calc_skewness, calc_kurtosis=calc_kurtosis,
                             calc_contrast=calc_contrast, calc_dissimilarity=calc_dissimilarity,
                             calc_homogeneity=calc_homogeneity, calc_ASM=calc_ASM, calc_energy=calc_energy,
                             calc_correlation=calc_correlation, start_label=first, ctx=ctx)
    return Segments(labels, objects, method, **kwargs)
