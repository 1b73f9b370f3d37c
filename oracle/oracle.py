"""ctypes front-end of the CPU parity oracle.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; ``obia_amd`` (the product) never does.  See ``obia_oracle.c`` for what each
function restates (reference file:line) and how it is pinned (scikit-image 0.18.3 golden vectors
under ``tests/golden/``).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libobia_oracle.so")
_lib = None

_i64 = ctypes.c_int64
_p = ctypes.c_void_p


def build(force=False):
    """Compile libobia_oracle.so with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "obia_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libobia_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.obia_oracle_grid_centroids.restype = _i64
        _lib.obia_oracle_masked_grid_centroids.restype = _i64
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_p)


def regular_grid(H, W, n_points):
    """(start_y, step_y, start_x, step_x); step 0 == slice(None).  util/_regular_grid.py:61-83."""
    out = np.zeros(4, np.int64)
    lib().obia_oracle_regular_grid(_i64(H), _i64(W), _i64(n_points), _ptr(out))
    return tuple(int(v) for v in out)


def grid_centroids(H, W, n_segments):
    steps = np.zeros(2, np.float64)
    K = lib().obia_oracle_grid_centroids(_i64(H), _i64(W), _i64(n_segments), None, _ptr(steps))
    yx = np.zeros((K, 2), np.int64)
    lib().obia_oracle_grid_centroids(_i64(H), _i64(W), _i64(n_segments), _ptr(yx), _ptr(steps))
    return yx, steps


def masked_grid_centroids(mask, n_segments):
    mask = np.ascontiguousarray(mask, np.uint8)
    H, W = mask.shape
    steps = np.zeros(2, np.float64)
    K = lib().obia_oracle_masked_grid_centroids(_ptr(mask), _i64(H), _i64(W), _i64(n_segments), None, _ptr(steps))
    yx = np.zeros((K, 2), np.int64)
    if K:
        lib().obia_oracle_masked_grid_centroids(_ptr(mask), _i64(H), _i64(W), _i64(n_segments), _ptr(yx), _ptr(steps))
    return yx, steps


def set_sum_mode(mode):
    """0: the reference's sequential float32 centroid sums (default); 1: the HIP path's integer sums (obia_oracle.c: g_sum_mode)."""
    lib().obia_oracle_set_sum_mode(ctypes.c_int(int(mode)))


def normalize(img):
    """obia normalize_band on every band (segment_boundaries.py:11-16,32-33); returns a copy."""
    out = np.ascontiguousarray(img, np.float32).copy()
    H, W, C = out.shape
    lib().obia_oracle_normalize(_ptr(out), _i64(H * W), ctypes.c_int(C))
    return out


def rgb2lab(rgb):
    rgb = np.ascontiguousarray(rgb, np.float32)
    out = np.empty_like(rgb)
    lib().obia_oracle_rgb2lab_f32(_ptr(rgb), _ptr(out), _i64(rgb.shape[0] * rgb.shape[1]))
    return out


# ---- Gaussian pre-smoothing: slic(..., sigma=...) -----------------------------------------------------------------------
# slic_superpixels.py (0.18.3): a scalar sigma becomes [s, s, s] / spacing, a sequence is taken as (z, y, x); when any entry is
# positive  `image = ndi.gaussian_filter(image, list(sigma) + [0])`  on the (1, H, W, C) image -- after the Lab conversion, before
# `* 1/compactness`.  scipy.ndimage.gaussian_filter (filters.py): one correlate1d per axis with sigma > 1e-15, in axis order, each
# pass written in the image's dtype (float32 here); weights = exp(-0.5 / sigma^2 * x^2) over x = -r..r, r = int(4 sigma + 0.5),
# divided by their NumPy sum (pairwise summation).  correlate1d (ni_filters.c, symmetric weights, mode 'reflect' = d c b a | a b c d |
# d c b a, repeated): the line is read as double,  tmp = line[i] * w[0];  for j = r .. 1: tmp += (line[i - j] + line[i + j]) * w[j].
# Pinned bit-exactly on scipy's output (tests/golden/sigma*.npz hold `smoothed`).
def _pairwise_sum(a):
    n = len(a)
    if n < 8:
        res = 0.0
        for v in a:
            res += float(v)
        return res
    if n <= 128:
        r = [float(a[j]) for j in range(8)]
        i = 8
        while i < n - (n % 8):
            for j in range(8):
                r[j] += float(a[i + j])
            i += 8
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        while i < n:
            res += float(a[i])
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return _pairwise_sum(a[:n2]) + _pairwise_sum(a[n2:])


def gaussian_weights_f64(sigma, truncate=4.0):
    """quickshift() hands scipy the caller's Python number: everything in float64 (scipy _gaussian_kernel1d)."""
    sd = float(sigma)
    lw = int(truncate * sd + 0.5)
    x = np.arange(-lw, lw + 1)
    phi = np.exp(-0.5 / (sd * sd) * x ** 2)
    return phi / _pairwise_sum(phi), lw


def gaussian_weights(sigma, truncate=4.0):
    # slic() hands scipy its sigmas as float32 scalars (the image's dtype); scipy 1.7.1 with NumPy 1.26 (the goldens' versions)
    # then forms `sigma2 = sigma * sigma` in FLOAT32 and everything after it in float64 (`sd = float(sigma)` for the radius,
    # `-0.5 / sigma2` a Python float over a NumPy scalar).  (NumPy >= 2 keeps `-0.5 / sigma2` in float32 as well: the weights of a
    # sigma whose square is not a float32 then differ from these by 1e-8 relative -- the smoothed image by an ulp here and there.)
    s32 = np.float32(sigma)
    sd = float(s32)
    lw = int(truncate * sd + 0.5)
    sigma2 = float(np.float32(s32 * s32))
    x = np.arange(-lw, lw + 1)
    phi = np.exp(-0.5 / sigma2 * x ** 2)
    return phi / _pairwise_sum(phi), lw


def _correlate1d_reflect(a, w, lw, axis):
    a = np.moveaxis(a, axis, -1)
    n = a.shape[-1]
    idx = np.mod(np.arange(-lw, n + lw), 2 * n)
    idx = np.where(idx >= n, 2 * n - 1 - idx, idx)
    ext = a[..., idx].astype(np.float64)
    tmp = ext[..., lw:lw + n] * w[lw]
    for jj in range(-lw, 0):
        tmp = tmp + (ext[..., lw + jj:lw + jj + n] + ext[..., lw - jj:lw - jj + n]) * w[lw + jj]
    return np.moveaxis(tmp.astype(a.dtype), -1, axis)


def sigma_zyx(sigma, spacing=None):
    """slic()'s reading of its `sigma` argument (slic_superpixels.py): the values live in the image's dtype (float32 here: 7.3 becomes
    7.30000019...); a number is [s, s, s] DIVIDED by the spacing, a sequence is taken as (z, y, x) as it is."""
    sp = np.ones(3, np.float32) if spacing is None else np.ascontiguousarray(spacing, dtype=np.float32)
    if np.isscalar(sigma):
        s = np.array([sigma, sigma, sigma], dtype=np.float32)
        s /= sp
    else:
        s = np.array(sigma, dtype=np.float32)
        if s.shape != (3,):
            raise ValueError("sigma: a number or a (z, y, x) sequence")
    return [float(v) for v in s]


def gaussian_filter_zyx(img_hwc, sigma, spacing=None):
    out = np.ascontiguousarray(img_hwc, np.float32)[None]           # (1, H, W, C): the depth axis has one plane and is filtered too
    for ax, s in enumerate(sigma_zyx(sigma, spacing)):
        if s > 1e-15:
            w, lw = gaussian_weights(s)
            out = _correlate1d_reflect(out, w, lw, ax)
    return np.ascontiguousarray(out[0])


def slic_core(image_scaled, segments, step, max_iter=10, mask=None, slic_zero=False,
              ignore_color=False, start_label=1):
    """_slic_cython; `segments` (K,2+C) float32 is updated in place. Returns labels (H,W) int64."""
    image_scaled = np.ascontiguousarray(image_scaled, np.float32)
    H, W, C = image_scaled.shape
    assert segments.dtype == np.float32 and segments.flags.c_contiguous and segments.shape[1] == 2 + C
    if mask is not None:
        mask = np.ascontiguousarray(mask, np.uint8)
    labels = np.empty((H, W), np.int64)
    rc = lib().obia_oracle_slic_core(_ptr(image_scaled), _ptr(mask), _ptr(segments), _i64(H), _i64(W),
                                     ctypes.c_int(C), _i64(segments.shape[0]), ctypes.c_float(step),
                                     ctypes.c_int(max_iter), ctypes.c_int(bool(slic_zero)),
                                     ctypes.c_int(bool(ignore_color)), ctypes.c_int(start_label), _ptr(labels))
    if rc:
        raise RuntimeError(f"oracle slic_core rc={rc}")
    return labels


def enforce_connectivity(labels, min_size, max_size, start_label=1):
    labels = np.ascontiguousarray(labels, np.int64)
    H, W = labels.shape
    out = np.empty_like(labels)
    rc = lib().obia_oracle_enforce_connectivity(_ptr(labels), _i64(H), _i64(W), _i64(min_size), _i64(max_size),
                                                ctypes.c_int(start_label), _ptr(out))
    if rc:
        raise RuntimeError(f"oracle enforce_connectivity rc={rc}")
    return out


def slic(image, n_segments=100, compactness=10.0, max_iter=10, convert2lab=None,
         enforce_connectivity=True, min_size_factor=0.5, max_size_factor=3, slic_zero=False,
         start_label=1, mask=None, seeds_yx=None, seed_steps=None, return_all=False, sigma=0, spacing=None):
    """skimage.segmentation.slic for a (H,W,C) float32 image (slic_superpixels.py:107-333).

    With ``mask`` and no ``seeds_yx`` the build's masked-grid seeding rule is used (see
    obia_oracle.c); with ``seeds_yx`` (K,2 float64) + ``seed_steps`` the given seeds are used,
    which is how the maskSLIC path is pinned on scikit-image's own seeds.
    """
    image = np.ascontiguousarray(image, np.float32)
    if image.ndim == 2:
        image = image[..., None]
    H, W, C = image.shape
    if mask is not None:
        mask = np.ascontiguousarray(mask, np.uint8)
    c2l = -1 if convert2lab is None else int(bool(convert2lab))
    if any(v > 0 for v in sigma_zyx(sigma, spacing)):   # smoothing sits between the Lab conversion and the scaling: Lab here, the rest in C
        if C == 3 and c2l != 0:
            image = rgb2lab(image)
        image = gaussian_filter_zyx(image, sigma, spacing)
        c2l = 0
    sp_yx = None if spacing is None else np.ascontiguousarray([float(np.float32(spacing[1])), float(np.float32(spacing[2]))], np.float64)
    labels = np.empty((H, W), np.int64)
    pre = np.empty((H, W), np.int64)
    if seeds_yx is not None:
        seeds_yx = np.ascontiguousarray(seeds_yx, np.float64)
        seed_steps = np.ascontiguousarray(seed_steps, np.float64)
        kmax = seeds_yx.shape[0]
    else:
        kmax = H * W
        if mask is None:
            kmax = grid_centroids(H, W, n_segments)[0].shape[0]
        else:
            kmax = max(1, masked_grid_centroids(mask, n_segments)[0].shape[0])
    cent = np.zeros((kmax, 2 + C), np.float32)
    K = _i64(0)
    rc = lib().obia_oracle_slic_sp(_ptr(image), _ptr(mask), _i64(H), _i64(W), ctypes.c_int(C), _i64(n_segments),
                                ctypes.c_double(compactness), ctypes.c_int(max_iter), ctypes.c_int(c2l),
                                ctypes.c_int(bool(enforce_connectivity)), ctypes.c_double(min_size_factor),
                                ctypes.c_double(max_size_factor), ctypes.c_int(bool(slic_zero)),
                                ctypes.c_int(start_label), _ptr(seeds_yx),
                                _i64(0 if seeds_yx is None else seeds_yx.shape[0]), _ptr(seed_steps),
                                _ptr(labels), _ptr(pre), _ptr(cent), ctypes.byref(K), _ptr(sp_yx))
    if rc:
        raise ValueError(f"oracle slic rc={rc}")
    if return_all:
        return labels, pre, cent[:K.value]
    return labels


def zonal_stats_numpy(raw, labels, bands=None, start_label=1, n_labels=None):
    """Bit-faithful restatement of calculate_spectral_stats (segment_statistics.py:143-172) under
    the equivalence of SURVEY.md 3.3: np.mean / np.var / np.min / np.max over the float32 pixels
    of each label.  Returns dict of arrays (n_labels, n_bands); NaN for empty labels."""
    raw = np.asarray(raw)
    H, W, C = raw.shape
    if bands is None:
        bands = list(range(C))
    lab = np.asarray(labels).ravel()
    if n_labels is None:
        n_labels = int(lab.max()) - start_label + 1 if lab.size else 0
    order = np.argsort(lab, kind="stable")
    sl = lab[order]
    out = {k: np.full((n_labels, len(bands)), np.nan, np.float64)
           for k in ("mean", "variance", "min", "max", "skewness", "kurtosis")}
    cnt = np.zeros(n_labels, np.int64)
    flat = raw.reshape(-1, C)
    lo = np.searchsorted(sl, np.arange(start_label, start_label + n_labels), "left")
    hi = np.searchsorted(sl, np.arange(start_label, start_label + n_labels), "right")
    for i in range(n_labels):
        idx = order[lo[i]:hi[i]]
        cnt[i] = idx.size
        if idx.size == 0:
            continue
        for j, b in enumerate(bands):
            v = flat[idx, b]
            v = v[~np.isnan(v)]
            if v.size == 0:
                continue
            out["mean"][i, j] = np.mean(v)
            out["variance"][i, j] = np.var(v)
            out["min"][i, j] = np.min(v)
            out["max"][i, j] = np.max(v)
            out["skewness"][i, j], out["kurtosis"][i, j] = skew_kurtosis(v)
    out["count"] = cnt
    return out


def skew_kurtosis(v):
    """scipy.stats.skew(v) and scipy.stats.kurtosis(v) with their defaults (bias=True, fisher=True), as
    calculate_spectral_stats calls them (segment_statistics.py:173-175).  scipy is a dependency of the reference
    (pyproject.toml: scipy>=1.14.1), not part of /root/reference; this restates its published algorithm
    (scipy/stats/_stats_py.py, 1.15.3): moments about the mean in the dtype of the data,
    m_k = mean((v - mean)**k);  skew = m3 / m2**1.5;  kurtosis = m4 / m2**2 - 3;  NaN where
    m2 <= (eps * mean)**2 (nearly constant data).  Pinned by tests/golden/moments_*.npz (generated by importing scipy)."""
    v = np.asarray(v)
    mean = v.mean()
    d = v - mean
    m2 = np.mean(d ** 2)
    m3 = np.mean(d ** 3)
    m4 = np.mean(d ** 4)
    eps = np.finfo(m2.dtype).eps
    with np.errstate(all="ignore"):
        if m2 <= (eps * mean) ** 2:
            return np.nan, np.nan
        return m3 / m2 ** 1.5, m4 / m2 ** 2.0 - 3


def zonal_stats_c(raw, labels, bands=None, start_label=1, n_labels=None):
    """float64-accumulating C port (timing baseline; obia_oracle_zonal_stats)."""
    raw = np.ascontiguousarray(raw, np.float32)
    H, W, C = raw.shape
    labels = np.ascontiguousarray(labels, np.int64)
    if bands is None:
        bands = list(range(C))
    b = np.ascontiguousarray(bands, np.int32)
    if n_labels is None:
        n_labels = int(labels.max()) - start_label + 1
    cnt = np.zeros(n_labels, np.int64)
    mean = np.zeros((n_labels, len(b)), np.float64)
    var = np.zeros_like(mean)
    mn = np.zeros((n_labels, len(b)), np.float32)
    mx = np.zeros_like(mn)
    rc = lib().obia_oracle_zonal_stats(_ptr(raw), _ptr(labels), _i64(H * W), ctypes.c_int(C), _ptr(b),
                                       ctypes.c_int(len(b)), _i64(n_labels), ctypes.c_int(start_label),
                                       _ptr(cnt), _ptr(mean), _ptr(var), _ptr(mn), _ptr(mx))
    if rc:
        raise RuntimeError(f"oracle zonal rc={rc}")
    return {"count": cnt, "mean": mean, "variance": var, "min": mn, "max": mx}


def quickshift_smooth(image_f64, sigma):
    """_quickshift.py: `image = ndi.gaussian_filter(image, [sigma, sigma, 0])` on the float64 (H, W, C) image, after the Lab conversion
    and before `* ratio`."""
    out = np.ascontiguousarray(image_f64, np.float64)
    if float(sigma) > 1e-15:
        w, lw = gaussian_weights_f64(sigma)
        out = _correlate1d_reflect(out, w, lw, 0)
        out = _correlate1d_reflect(out, w, lw, 1)
    return np.ascontiguousarray(out)


def quickshift_core(image_f64, noise, kernel_size, max_dist):
    image_f64 = np.ascontiguousarray(image_f64, np.float64)
    noise = np.ascontiguousarray(noise, np.float64)
    H, W, C = image_f64.shape
    out = np.empty((H, W), np.int64)
    rc = lib().obia_oracle_quickshift_core(_ptr(image_f64), _ptr(noise), _i64(H), _i64(W), ctypes.c_int(C),
                                           ctypes.c_double(kernel_size), ctypes.c_double(max_dist), _ptr(out))
    if rc:
        raise RuntimeError(f"oracle quickshift rc={rc}")
    return out
