#!/opt/conda/bin/python3.9
"""Golden vectors for `slic(..., sigma=...)` from scikit-image 0.18.3 / SciPy 1.7.1 (the Gaussian pre-smoothing of
slic_superpixels.py: `image = ndi.gaussian_filter(image, sigma + [0])`, applied after the Lab conversion and before `* 1/compactness`).

Run ONLY in the build container:   /opt/conda/bin/python3.9 tests/golden/gen_goldens_sigma.py
Fixtures hold DATA only (inputs as uint16 digital numbers, parameters, the labels and the smoothed image scikit-image / SciPy produced).
"""
import os
import warnings

import numpy as np

warnings.filterwarnings("ignore")
import scipy.ndimage as ndi  # noqa: E402
import skimage  # noqa: E402
from skimage.color import rgb2lab  # noqa: E402
from skimage.segmentation import slic, quickshift  # noqa: E402
from skimage.segmentation import slic_superpixels as _ss  # noqa: E402

from gen_goldens import obia_normalize, synth_dn  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def case(name, raw, params, sigma, mask=None, spacing=None):
    raw_f32 = raw.astype(np.float32)
    img = obia_normalize(raw_f32)
    kw = dict(n_segments=params["n_segments"], compactness=params["compactness"], max_iter=10, sigma=sigma, multichannel=True,
              convert2lab=params.get("convert2lab", None), start_label=1)
    if spacing is not None:
        kw["spacing"] = list(spacing)
    seeds = {}
    if mask is not None:   # (maskSLIC is pinned on scikit-image's own seeds, as in gen_goldens.py)
        m3 = np.ascontiguousarray(mask[np.newaxis, ...], dtype=bool).view("uint8")
        cent, steps = _ss._get_mask_centroids(m3, params["n_segments"], True)
        seeds = dict(seeds_yx=cent[:, 1:3].astype(np.float64), seed_steps=np.asarray(steps[1:3], np.float64),
                     seed_steps_all=np.asarray(steps, np.float64))
        kw["mask"] = mask
    pre = slic(img, enforce_connectivity=False, **kw)
    fin = slic(img, enforce_connectivity=True, **kw)
    # the smoothed image as slic() forms it: sigma in the image's dtype (float32), a scalar divided by the spacing, a list taken as is
    sp = np.ones(3, np.float32) if spacing is None else np.ascontiguousarray(spacing, dtype=np.float32)
    if np.isscalar(sigma):
        sg = np.array([sigma, sigma, sigma], dtype=np.float32)
        sg /= sp
    else:
        sg = np.array(sigma, dtype=np.float32)
    sig = [float(v) for v in sg]
    feat = rgb2lab(img) if (img.shape[2] == 3 and params.get("convert2lab", None) in (None, True)) else img
    smooth = ndi.gaussian_filter(feat[np.newaxis].astype(np.float32), list(sg) + [0])[0] if (sg > 0).any() else feat
    out = dict(raw=raw, labels_pre=pre.astype(np.int32), labels=fin.astype(np.int32), smoothed=smooth.astype(np.float32),
               sigma_zyx=np.asarray(sig, np.float64), sigma_arg=np.asarray(sigma, np.float64),
               spacing_zyx=np.asarray(sp, np.float64), params=np.array(repr(params)), skimage_version=np.array(skimage.__version__))
    if mask is not None:
        out["mask"] = mask.astype(np.uint8)
        out.update(seeds)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, raw.shape, "sigma", sig, "K_pre", len(np.unique(pre)), "N_final", len(np.unique(fin)))


def main():
    case("sigma1_96x112x4", synth_dn(96, 112, 4, seed=21), dict(n_segments=60, compactness=0.3), 1.0)
    case("sigma2_lab_100x120x3", synth_dn(100, 120, 3, seed=22), dict(n_segments=50, compactness=10.0), 2.0)
    case("sigma_list_80x90x8", synth_dn(80, 90, 8, seed=23), dict(n_segments=40, compactness=0.5), [0.0, 1.5, 0.7])
    case("sigma_wide_33x41x4", synth_dn(33, 41, 4, seed=24), dict(n_segments=12, compactness=1.0), 12.0)   # radius 48 > the raster: repeated reflections
    H, W = 96, 128
    yy, xx = np.mgrid[0:H, 0:W]
    mask = ((yy - 48) ** 2 + (xx - 64) ** 2 < 44 ** 2)
    mask[20:36, 50:70] = False
    case("sigma_mask_96x128x4", synth_dn(H, W, 4, seed=25), dict(n_segments=40, compactness=0.4), 1.5, mask=mask)
    case("sigma_f32_70x80x4", synth_dn(70, 80, 4, seed=26), dict(n_segments=30, compactness=0.5), 7.3)      # 7.3 is not a float32: slic() rounds it
    # spacing: anisotropic pixels (the row / column differences are scaled before they are squared; a scalar sigma is divided by it)
    case("spacing_90x100x4", synth_dn(90, 100, 4, seed=27), dict(n_segments=45, compactness=0.6), 0, spacing=[1.0, 2.0, 0.5])
    case("spacing_sigma_90x100x4", synth_dn(90, 100, 4, seed=28), dict(n_segments=45, compactness=0.6), 1.2, spacing=[3.0, 1.5, 0.75])
    case("spacing_mask_96x128x4", synth_dn(H, W, 4, seed=29), dict(n_segments=40, compactness=0.4), 0, mask=mask, spacing=[1.0, 0.6, 1.7])


def quickshift_cases():
    qs = {}
    for i, (H, W, C, ks, md, sg, lab) in enumerate([(36, 44, 3, 3, 10, 1.5, True), (40, 52, 4, 2, 6, 0.8, False), (30, 30, 1, 5, 10, 2.5, False)]):
        raw = synth_dn(H, W, C, seed=40 + i)
        img64 = obia_normalize(raw.astype(np.float32)).astype(np.float64)   # (0.18.3's kernel is float64-only, as in gen_goldens.py)
        out = quickshift(img64, ratio=0.7, kernel_size=ks, max_dist=md, sigma=sg, convert2lab=lab, random_seed=42)
        feat = rgb2lab(img64) if lab else img64
        qs[f"raw{i}"] = raw
        qs[f"feat{i}"] = feat                                                  # the image the smoothing starts from
        qs[f"smoothed{i}"] = ndi.gaussian_filter(feat, [sg, sg, 0])
        qs[f"labels{i}"] = out.astype(np.int32)
        qs[f"par{i}"] = np.array([ks, md, sg, 0.7, float(lab)], np.float64)
    np.savez_compressed(os.path.join(HERE, "quickshift_sigma.npz"), **qs)
    print("quickshift_sigma", 3, "cases")


if __name__ == "__main__":
    main()
    quickshift_cases()
