"""Host-side mirror of ``obia.utils.tiling.create_tiled_segments`` (obia/utils/tiling.py:62-291).

Same parameter names (``tile_size``, ``buffer``, ``crown_radius``, ``input_mask``, ``method`` -- only
"slic", else ValueError, tiling.py:76-77 -- and ``**kwargs`` forwarded to SLIC).  The array-level variant
returns the global label raster with ids 1..N (tiling.py:289-290) instead of writing ``segments.gpkg``:
polygonisation is the stage after this path.  The tile loops, the seam logic and the
per-tile SLIC all run in libobia_hip.so (tiling.hip); there is no CPU path.
"""
import ctypes

import numpy as np

from . import _lib
from .segmentation import make_params, _is_torch, _SLIC_KWARGS

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


def _open_raster(path):
    """GeoTIFF -> ((H,W,C) float32, pixel_width, pixel_height) via GDAL, as _create_tile does band by band
    (tiling.py:37-59).  GDAL is optional: the array-level entry needs no geo stack."""
    try:
        from osgeo import gdal
    except Exception as e:  # pragma: no cover
        raise ImportError("reading a raster path needs GDAL (osgeo); pass the (H,W,C) array instead") from e
    ds = gdal.Open(path)
    if not ds:
        raise ValueError(f"Unable to open {path}")
    gt = ds.GetGeoTransform()
    arr = np.empty((ds.RasterYSize, ds.RasterXSize, ds.RasterCount), np.float32)
    for b in range(ds.RasterCount):
        arr[:, :, b] = ds.GetRasterBand(b + 1).ReadAsArray()
    return arr, abs(gt[1]), abs(gt[5])


def write_segments_gpkg(labels, output_dir, affine_transformation=None, crs=None, ctx=None):
    """``all_segments.to_file(output_dir/"segments.gpkg", driver="GPKG")`` (tiling.py:289-291) for a label raster with ids
    1..N: polygons from the GPU polygoniser, columns ``geom`` / ``segment_id``, written with the standard library
    (obia_amd.geopackage).  Returns the path."""
    import os
    from .geopackage import write_geopackage
    from .polygons import polygonize
    from .segmentation import _epsg_of
    os.makedirs(output_dir, exist_ok=True)
    pt = polygonize(labels, affine_transformation=affine_transformation, start_label=1, ctx=ctx)
    path = os.path.join(output_dir, "segments.gpkg")
    write_geopackage(path, pt.wkb(), {"segment_id": np.arange(1, len(pt) + 1)}, table="segments", srs_epsg=_epsg_of(crs))
    return path


def create_tiled_segments(input_raster, output_dir=None, input_mask=None, method="slic", tile_size=200, buffer=30,
                          crown_radius=5, pixel_size=None, white_order="raster", affine_transformation=None, crs=None, ctx=None,
                          **kwargs):
    """Tiled SLIC over a large raster.

    input_raster : path (GDAL), object with ``img_data``, (H,W,C) NumPy array, or CUDA tensor.
    input_mask   : path / (H,W) array, optional; 0 = excluded.  ``None`` means every pixel is valid (the
        reference only works with a mask, tiling.py:84 ``# todo: mask should be optional``).
    pixel_size   : (width, height) of a pixel in map units; taken from the geotransform for paths,
        default (1, 1).  Enters the crown rule n = round(valid_px * pixel_area / (pi * crown_radius^2))
        (tiling.py:126-135) and the side of the masked corner squares (buffer/2 map units, tiling.py:189).
    white_order  : "raster" (the reference's order of white tiles) or "parity" (even tile rows, then odd ones:
        the order of the multi-GPU driver, obia_amd.distributed).
    output_dir   : when given, ``segments.gpkg`` is written there like the reference does (tiling.py:289-291; columns
        geometry + segment_id), from either entry (NumPy array or CUDA tensor).  ``affine_transformation`` ([a, b, d, e,
        xoff, yoff], obia ``Image.affine_transformation``; taken from the geotransform for paths, default: pixel size and
        a north-up origin at (0, 0)) and ``crs`` ("EPSG:xxxx") place the polygons.
    kwargs       : SLIC keyword arguments (scikit-image names).  ``n_segments`` (which the reference
        cannot accept: duplicate keyword TypeError, tiling.py:126,137-143) is taken per full tile and
        scaled by the tile's valid area.
    Returns (labels (H,W) int32 with ids 1..N and 0 where no segment, N).
    """
    if method != "slic":
        raise ValueError("Currently, only the 'slic' method is supported for segmentation.")
    unknown = [k for k in kwargs if k not in _SLIC_KWARGS]
    if unknown:
        raise TypeError(f"slic() got an unexpected keyword argument '{unknown[0]}'")
    if kwargs.get("mask") is not None:
        raise TypeError("create_segments() got multiple values for keyword argument 'mask'")
    pw = ph = None
    if isinstance(input_raster, str):
        img, pw, ph = _open_raster(input_raster)
    else:
        img = input_raster.img_data if hasattr(input_raster, "img_data") else input_raster
        if affine_transformation is None:
            affine_transformation = getattr(input_raster, "affine_transformation", None)
        if crs is None:
            crs = getattr(input_raster, "crs", None)
    if isinstance(input_mask, str):
        m, _, _ = _open_raster(input_mask)
        input_mask = m[:, :, 0] != 0
    if pixel_size is not None:
        pw, ph = float(pixel_size[0]), float(pixel_size[1])
    if pw is None:
        pw = ph = 1.0
    n_seg = kwargs.get("n_segments", None)
    params = make_params(n_segments=0 if n_seg is None else n_seg, compactness=kwargs.get("compactness", 10.0),
                         max_num_iter=kwargs.get("max_num_iter", kwargs.get("max_iter", 10) or 10),
                         convert2lab=kwargs.get("convert2lab", None),
                         enforce_connectivity=kwargs.get("enforce_connectivity", True),
                         min_size_factor=kwargs.get("min_size_factor", 0.5),
                         max_size_factor=kwargs.get("max_size_factor", 3), slic_zero=kwargs.get("slic_zero", False),
                         start_label=1, normalize_bands=True, exit_on_fixed_point=kwargs.get("exit_on_fixed_point", False),
                         sigma=kwargs.get("sigma", 0), spacing=kwargs.get("spacing"))
    if white_order not in ("raster", "parity"):
        raise ValueError("white_order must be 'raster' or 'parity'")
    tp = _lib.TilingParams()
    tp.tile_size, tp.buffer = int(tile_size), int(buffer)
    tp.white_order = 1 if white_order == "parity" else 0
    tp.crown_radius, tp.pixel_width, tp.pixel_height = float(crown_radius), float(pw), float(ph)
    lib = _lib.load()
    n_out = ctypes.c_int64(0)
    if _is_torch(img):
        if not img.is_cuda:
            raise ValueError("torch inputs must live on the GPU")
        x = img.to(torch.float32).contiguous()
        H, W, C = x.shape
        m = None
        if input_mask is not None:
            m = _lib.mask_bytes(input_mask, x.device)
            if tuple(m.shape) != (H, W):
                raise ValueError("image and mask should have the same shape.")
        dev = x.device.index or 0
        c = ctx or _lib.default_context(dev)
        torch.cuda.current_stream(dev).synchronize()
        out = torch.empty((H, W), dtype=torch.int32, device=x.device)
        _lib.check(lib.obia_tiled_slic_f32_dev(c.handle, x.data_ptr(), m.data_ptr() if m is not None else None, H, W, C,
                                               ctypes.byref(tp), ctypes.byref(params), out.data_ptr(), ctypes.byref(n_out)))
        if output_dir is not None:
            write_segments_gpkg(out, output_dir, affine_transformation or [pw, 0.0, 0.0, -ph, 0.0, 0.0], crs, ctx=c)
        return out, int(n_out.value)
    x = np.ascontiguousarray(img, dtype=np.float32)
    if x.ndim != 3:
        raise ValueError("raster must be (H,W,C)")
    H, W, C = x.shape
    m = None
    if input_mask is not None:
        m = np.ascontiguousarray(np.asarray(input_mask) != 0, dtype=np.uint8)
        if m.shape != (H, W):
            raise ValueError("image and mask should have the same shape.")
    c = ctx or _lib.default_context(0)
    out = np.empty((H, W), np.int32)
    _lib.check(lib.obia_tiled_slic_f32(c.handle, _lib.np_ptr(x), _lib.np_ptr(m), H, W, C, ctypes.byref(tp),
                                       ctypes.byref(params), _lib.np_ptr(out), ctypes.byref(n_out)))
    if output_dir is not None:
        write_segments_gpkg(out, output_dir, affine_transformation or [pw, 0.0, 0.0, -ph, 0.0, 0.0], crs, ctx=c)
    return out, int(n_out.value)
