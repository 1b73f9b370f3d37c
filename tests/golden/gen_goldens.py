#!/opt/conda/bin/python3.9
"""Generate golden vectors for the obia hot path from scikit-image 0.18.3.

Run ONLY in the build container:   /opt/conda/bin/python3.9 tests/golden/gen_goldens.py
It imports scikit-image (the third-party library that holds the arithmetic obia calls at
obia/segmentation/segment_boundaries.py:48-51) and writes small .npz fixtures next to this file.
Fixtures hold DATA only: inputs (uint16 digital numbers or float32), parameters, and the outputs
scikit-image / NumPy produced.  Nothing from scikit-image or the reference is copied.

Keyword names are those of 0.18.3 (max_iter, multichannel, random_seed); the reference pins
`scikit-image>=0.23.2` -- see SURVEY.md 8c for the version deltas the cases below neutralise
(explicit start_label, explicit max_iter, inputs already in [0,1]).
"""
import os
import warnings

import numpy as np

warnings.filterwarnings("ignore")
from skimage.segmentation import slic, quickshift  # noqa: E402
from skimage.segmentation import slic_superpixels as _ss  # noqa: E402
from skimage.segmentation._slic import _enforce_label_connectivity_cython  # noqa: E402
from skimage.color import rgb2lab  # noqa: E402
import skimage  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def synth_dn(H, W, C, seed=0):
    """BASELINE.md 3 generator, quantised to uint16 digital numbers so the fixture is exact."""
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    out = np.empty((H, W, C), np.uint16)
    for c in range(C):
        b = 400.0 * np.sin(xx / (11 + 3 * c)) * np.cos(yy / (13 + 2 * c)) + 1000 + 50 * c + rs.normal(0, 20, (H, W))
        out[:, :, c] = np.clip(np.rint(b), 0, 65535).astype(np.uint16)
    return out


def obia_normalize(raw_f32):
    """obia/segmentation/segment_boundaries.py:16,32-33 in NumPy (float32, every band)."""
    img = raw_f32.copy()
    for i in range(img.shape[2]):
        b = img[:, :, i]
        img[:, :, i] = (b - np.min(b)) / (np.max(b) - np.min(b))
    return img


def zonal(raw_f32, labels, start_label=1):
    """np.mean/var/min/max per label on float32 pixels (segment_statistics.py:165-172)."""
    n = int(labels.max()) - start_label + 1
    C = raw_f32.shape[2]
    mean = np.full((n, C), np.nan, np.float64)
    var = np.full((n, C), np.nan, np.float64)
    mn = np.full((n, C), np.nan, np.float64)
    mx = np.full((n, C), np.nan, np.float64)
    cnt = np.zeros(n, np.int64)
    flat = raw_f32.reshape(-1, C)
    lab = labels.ravel()
    order = np.argsort(lab, kind="stable")
    sl = lab[order]
    for i in range(n):
        lo, hi = np.searchsorted(sl, [i + start_label, i + start_label + 1])
        idx = order[lo:hi]
        cnt[i] = idx.size
        if idx.size == 0:
            continue
        for c in range(C):
            v = flat[idx, c]
            mean[i, c] = np.mean(v)
            var[i, c] = np.var(v)
            mn[i, c] = np.min(v)
            mx[i, c] = np.max(v)
    return dict(z_count=cnt, z_mean=mean, z_var=var, z_min=mn, z_max=mx)


def slic_case(name, raw, params, mask=None, store_lab=False, with_zonal=True):
    raw_f32 = raw.astype(np.float32)
    img = obia_normalize(raw_f32)
    kw = dict(n_segments=params["n_segments"], compactness=params["compactness"],
              max_iter=params.get("max_iter", 10), sigma=0, multichannel=True,
              convert2lab=params.get("convert2lab", None),
              min_size_factor=params.get("min_size_factor", 0.5),
              max_size_factor=params.get("max_size_factor", 3),
              slic_zero=params.get("slic_zero", False), start_label=params.get("start_label", 1))
    out = {}
    if mask is not None:
        m3 = np.ascontiguousarray(mask[np.newaxis, ...], dtype=bool).view("uint8")
        cent, steps = _ss._get_mask_centroids(m3, params["n_segments"], True)
        out["seeds_yx"] = cent[:, 1:3].astype(np.float64)
        out["seed_steps"] = np.asarray(steps[1:3], np.float64)
        out["seed_steps_all"] = np.asarray(steps, np.float64)
        out["mask"] = mask.astype(np.uint8)
        kw["mask"] = mask
    pre = slic(img, enforce_connectivity=False, **kw)
    fin = slic(img, enforce_connectivity=True, **kw)
    out.update(raw=raw, labels_pre=pre.astype(np.int32), labels=fin.astype(np.int32),
               params=np.array(repr(params)), skimage_version=np.array(skimage.__version__))
    if store_lab:
        out["lab"] = rgb2lab(img).astype(np.float32)
    if with_zonal:
        out.update(zonal(raw_f32, fin, params.get("start_label", 1)))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, raw.shape, "K_pre", len(np.unique(pre)), "N_final", len(np.unique(fin)))


def main():
    # 1. quickstart notebook input (docs/examples/segmentation-quickstart.ipynb cells 2-4)
    h = w = 128
    yy, xx = np.mgrid[0:h, 0:w]
    q = np.stack([xx / w * 1000, yy / h * 1000, (np.sin(xx / 8) + np.cos(yy / 10)) * 200 + 500], axis=-1).astype(np.float32)
    slic_case("quickstart_128x128x3", q, dict(n_segments=200, compactness=8, start_label=1), store_lab=True)

    # 2. BASELINE config 1
    slic_case("c1_512x512x3", synth_dn(512, 512, 3), dict(n_segments=500, compactness=10))
    # 3. config-2 analogue (S=18), compactness 10 and the author's regime 0.25 (fragments -> merge path)
    slic_case("c2s_256x256x4_c10", synth_dn(256, 256, 4), dict(n_segments=200, compactness=10))
    slic_case("c2s_256x256x4_c025", synth_dn(256, 256, 4), dict(n_segments=200, compactness=0.25))
    slic_case("c2s_256x256x4_c005", synth_dn(256, 256, 4), dict(n_segments=200, compactness=0.05))
    # 4. config-3 tile analogue (8 bands, S=18)
    slic_case("c3s_384x384x8_c10", synth_dn(384, 384, 8), dict(n_segments=455, compactness=10))
    slic_case("c3s_384x384x8_c025", synth_dn(384, 384, 8), dict(n_segments=455, compactness=0.25))
    # 5. K != n_segments, ragged shapes, other parameters
    slic_case("k_ne_n_75x75x4", synth_dn(75, 75, 4, seed=3), dict(n_segments=33, compactness=1.0))
    slic_case("ragged_200x333x5", synth_dn(200, 333, 5, seed=4), dict(n_segments=150, compactness=1.0))
    slic_case("thin_7x400x4", synth_dn(7, 400, 4, seed=5), dict(n_segments=40, compactness=0.5))
    slic_case("iter3_100x120x4", synth_dn(100, 120, 4, seed=6), dict(n_segments=50, compactness=0.3, max_iter=3))
    slic_case("label0_100x120x4", synth_dn(100, 120, 4, seed=6), dict(n_segments=50, compactness=0.3, start_label=0))
    slic_case("sliczero_100x120x4", synth_dn(100, 120, 4, seed=7), dict(n_segments=50, compactness=0.3, slic_zero=True))
    slic_case("sizefac_150x150x4", synth_dn(150, 150, 4, seed=8),
              dict(n_segments=60, compactness=0.1, min_size_factor=0.9, max_size_factor=1.2))
    slic_case("nolab_96x96x3", synth_dn(96, 96, 3, seed=9), dict(n_segments=40, compactness=0.5, convert2lab=False))
    slic_case("onech_90x110x1", synth_dn(90, 110, 1, seed=10), dict(n_segments=45, compactness=0.2))
    slic_case("morethanpx_6x7x4", synth_dn(6, 7, 4, seed=11), dict(n_segments=100, compactness=1.0), with_zonal=False)
    # 6. maskSLIC, pinned on scikit-image's own seeds
    H, W = 128, 160
    yy, xx = np.mgrid[0:H, 0:W]
    mask = ((yy - 60) ** 2 + (xx - 80) ** 2 < 55 ** 2)
    mask[30:50, 70:90] = False
    slic_case("mask_128x160x4_c10", synth_dn(H, W, 4, seed=12), dict(n_segments=60, compactness=10), mask=mask)
    slic_case("mask_128x160x4_c025", synth_dn(H, W, 4, seed=12), dict(n_segments=60, compactness=0.25), mask=mask)
    slic_case("maskones_96x96x4", synth_dn(96, 96, 4, seed=13), dict(n_segments=30, compactness=1.0),
              mask=np.ones((96, 96), bool))

    # 7. connectivity enforcement, black-box on noisy label maps (splits and merges both exercised)
    rs = np.random.RandomState(21)
    cases = {}
    for i, (H, W, nlab, mn, mx) in enumerate([(40, 50, 6, 5, 60), (64, 64, 12, 10, 200), (30, 90, 4, 20, 100),
                                               (50, 50, 3, 3, 40), (33, 47, 9, 1, 15)]):
        base = (np.arange(H)[:, None] // max(1, H // 3)) * 3 + (np.arange(W)[None, :] // max(1, W // 3))
        noise = rs.randint(0, nlab, (H, W))
        lab = np.where(rs.rand(H, W) < 0.25, noise, base % nlab).astype(np.intp) + 1
        if i == 2:
            lab[rs.rand(H, W) < 0.1] = 0     # masked pixels (start_label - 1)
        out = _enforce_label_connectivity_cython(np.ascontiguousarray(lab[np.newaxis]), mn, mx, start_label=1)
        cases[f"in{i}"] = lab.astype(np.int32)
        cases[f"out{i}"] = np.asarray(out)[0].astype(np.int32)
        cases[f"par{i}"] = np.array([mn, mx], np.int64)
    np.savez_compressed(os.path.join(HERE, "connectivity_blackbox.npz"), **cases)
    print("connectivity_blackbox", len(cases) // 3, "cases")

    # 8. quickshift (config 5), small cases; Lab image stored so the checker starts at the kernel
    qs = {}
    for i, (H, W, ks, md) in enumerate([(32, 40, 2, 6), (36, 36, 3, 10), (40, 48, 5, 10)]):
        raw = synth_dn(H, W, 3, seed=30 + i)
        img = obia_normalize(raw.astype(np.float32))
        # 0.18.3's kernel is float64-only (float32 support came later); feed the float32-normalised
        # image widened to float64
        img64 = img.astype(np.float64)
        lab = quickshift(img64, ratio=1.0, kernel_size=ks, max_dist=md, sigma=0, convert2lab=True, random_seed=42)
        qs[f"raw{i}"] = raw
        qs[f"lab{i}"] = rgb2lab(img64)
        qs[f"labels{i}"] = lab.astype(np.int32)
        qs[f"par{i}"] = np.array([ks, md], np.float64)
    np.savez_compressed(os.path.join(HERE, "quickshift_small.npz"), **qs)
    print("quickshift_small", 3, "cases")


if __name__ == "__main__":
    main()
