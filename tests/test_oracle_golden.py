"""Pin the CPU oracle (oracle/obia_oracle.c) on the scikit-image 0.18.3 golden vectors.

scikit-image is the third-party library that holds the arithmetic obia calls
(obia/segmentation/segment_boundaries.py:48-51); fixtures come from tests/golden/gen_goldens.py.
Integer outputs are required BIT-EXACT wherever the input to the integer stage is identical;
3-band cases go through float32 rgb2lab (numpy SIMD pow/cbrt + BLAS) whose last-ulp behaviour a
restatement cannot reproduce, so they carry a stated pixel tolerance.
"""
import ast
import glob
import os

import numpy as np
import pytest

from tests.metrics import adjusted_rand_index, label_disagreement

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SLIC_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "*.npz"))
                    if not os.path.basename(p).startswith(("connectivity_", "quickshift_", "moments_", "glcm_", "sigma", "spacing")))
SIGMA_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "sigma*.npz")) + glob.glob(os.path.join(GOLD, "spacing*.npz")))


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    params = ast.literal_eval(str(z["params"]))
    return z, params


def run_oracle(oracle, z, params):
    raw = z["raw"].astype(np.float32)
    img = oracle.normalize(raw)
    kw = dict(n_segments=params["n_segments"], compactness=params["compactness"],
              max_iter=params.get("max_iter", 10), convert2lab=params.get("convert2lab", None),
              min_size_factor=params.get("min_size_factor", 0.5), max_size_factor=params.get("max_size_factor", 3),
              slic_zero=params.get("slic_zero", False), start_label=params.get("start_label", 1))
    if "mask" in z.files:
        kw.update(mask=z["mask"], seeds_yx=z["seeds_yx"], seed_steps=z["seed_steps"])
        # step = max(steps) over all three axes in the reference (depth axis included)
        kw["seed_steps"] = np.array([max(z["seed_steps_all"][0], z["seed_steps"][0]), z["seed_steps"][1]])
    return oracle.slic(img, return_all=True, **kw)


@pytest.mark.parametrize("name", SLIC_CASES)
def test_slic_labels_match_skimage(oracle, name):
    z, params = load(name)
    labels, pre, cent = run_oracle(oracle, z, params)
    three_band_lab = z["raw"].shape[2] == 3 and params.get("convert2lab", None) is not False
    if three_band_lab:
        # float32 Lab differs from numpy's in the last ulp on a few pixels -> a handful of label flips
        assert label_disagreement(pre, z["labels_pre"]) <= 2e-4
        assert adjusted_rand_index(labels, z["labels"]) >= 0.995
    else:
        assert np.array_equal(pre, z["labels_pre"]), f"pre-connectivity labels differ: {(pre != z['labels_pre']).sum()} px"
        assert np.array_equal(labels, z["labels"]), f"final labels differ: {(labels != z['labels']).sum()} px"


@pytest.mark.parametrize("name", SIGMA_CASES)
def test_slic_with_sigma_matches_skimage(oracle, name):
    """slic(..., sigma=..., spacing=...): the Gaussian pre-smoothing (scipy.ndimage.gaussian_filter between the Lab conversion and the
    scaling; a scalar sigma divided by the spacing, everything in float32 as the image) and the anisotropic distance term -- the
    smoothed image bit for bit (Lab rasters: within the Lab tolerance), the labels as in the other cases.
    Fixtures: tests/golden/gen_goldens_sigma.py (scikit-image 0.18.3, SciPy 1.7.1, NumPy 1.26)."""
    z, params = load(name)
    img = oracle.normalize(z["raw"].astype(np.float32))
    sig = float(z["sigma_arg"]) if z["sigma_arg"].shape == () else [float(v) for v in z["sigma_arg"]]
    sp = [float(v) for v in z["spacing_zyx"]]
    sp = None if sp == [1.0, 1.0, 1.0] else sp
    assert oracle.sigma_zyx(sig, sp) == [float(v) for v in z["sigma_zyx"]]
    lab_case = z["raw"].shape[2] == 3 and params.get("convert2lab", None) is not False
    sm = oracle.gaussian_filter_zyx(oracle.rgb2lab(img) if lab_case else img, sig, sp)
    if lab_case:
        assert np.abs(sm - z["smoothed"]).max() <= 2e-4
    else:
        assert np.array_equal(sm, z["smoothed"])
    kw = dict(n_segments=params["n_segments"], compactness=params["compactness"], sigma=sig, spacing=sp)
    if "mask" in z.files:
        kw.update(mask=z["mask"], seeds_yx=z["seeds_yx"], seed_steps=z["seed_steps"])
    labels, pre, _ = oracle.slic(img, return_all=True, **kw)
    if lab_case:
        assert label_disagreement(pre, z["labels_pre"]) <= 2e-4
        assert adjusted_rand_index(labels, z["labels"]) >= 0.995
    else:
        assert np.array_equal(pre, z["labels_pre"]) and np.array_equal(labels, z["labels"])


def test_gaussian_weights_sum_like_numpy(oracle):
    """the weights are normalised by NumPy's pairwise sum, restated in `_pairwise_sum` (the order of the additions decides the last bit)"""
    rs = np.random.RandomState(0)
    for n in (1, 5, 7, 8, 9, 17, 97, 128, 129, 200, 1000):
        a = rs.rand(n)
        assert oracle._pairwise_sum(a) == float(np.sum(a)), n


def test_integer_sum_mode_of_the_oracle(oracle):
    """`set_sum_mode(1)` swaps ONLY the centroid update's summation for the HIP path's integer sums (obia_oracle.c: g_sum_mode; used by
    tests/test_gpu_exact_sums.py to attribute the last differences).  At compactness 10 nothing is near a tie: the golden's labels come
    out unchanged; the centroids themselves differ in their last bits, which is the whole point."""
    z, params = load("c2s_256x256x4_c10")
    ref = run_oracle(oracle, z, params)
    oracle.set_sum_mode(1)
    try:
        alt = run_oracle(oracle, z, params)
    finally:
        oracle.set_sum_mode(0)
    assert np.array_equal(alt[0], z["labels"]) and np.array_equal(alt[1], z["labels_pre"])
    assert not np.array_equal(alt[2], ref[2]) and np.abs(alt[2] - ref[2]).max() <= 1e-4 * np.abs(ref[2]).max()


def test_rgb2lab_close_to_skimage(oracle):
    z, _ = load("quickstart_128x128x3")
    img = oracle.normalize(z["raw"].astype(np.float32))
    lab = oracle.rgb2lab(img)
    # L in [0,100], a/b in [-128,128]; float32 pow/cbrt/3x3 product: stated tolerance 2e-4 absolute
    assert np.abs(lab - z["lab"]).max() <= 2e-4


def test_connectivity_blackbox_bit_exact(oracle):
    z = np.load(os.path.join(GOLD, "connectivity_blackbox.npz"))
    n = len([k for k in z.files if k.startswith("in")])
    assert n >= 5
    for i in range(n):
        mn, mx = (int(v) for v in z[f"par{i}"])
        out = oracle.enforce_connectivity(z[f"in{i}"].astype(np.int64), mn, mx, start_label=1)
        assert np.array_equal(out, z[f"out{i}"]), f"case {i}"


def test_quickshift_small_bit_exact(oracle):
    z = np.load(os.path.join(GOLD, "quickshift_small.npz"))
    for i in range(3):
        ks, md = z[f"par{i}"]
        lab = z[f"lab{i}"]
        noise = np.random.RandomState(42).normal(scale=0.00001, size=lab.shape[:2])
        out = oracle.quickshift_core(lab * 1.0, noise, ks, md)
        assert np.array_equal(out, z[f"labels{i}"]), f"case {i}: {(out != z[f'labels{i}']).sum()} px differ"


def test_quickshift_with_sigma_bit_exact(oracle):
    """quickshift(..., sigma=...): `ndi.gaussian_filter(image, [sigma, sigma, 0])` on the float64 image after the Lab conversion,
    before `* ratio` (_quickshift.py) -- the smoothed float64 image to the last bit or two (the weights go through `exp`, whose last bit
    differs between the NumPy that made the fixture and the one that runs this test; a float64 result has no float32 rounding to hide
    that), the labels exact (gen_goldens_sigma.py)."""
    z = np.load(os.path.join(GOLD, "quickshift_sigma.npz"))
    for i in range(3):
        ks, md, sg, ratio, _lab = z[f"par{i}"]
        sm = oracle.quickshift_smooth(z[f"feat{i}"], float(sg))
        assert np.abs(sm - z[f"smoothed{i}"]).max() <= 4e-16 * np.abs(z[f"smoothed{i}"]).max(), f"case {i}: smoothed image differs"
        noise = np.random.RandomState(42).normal(scale=0.00001, size=sm.shape[:2])
        out = oracle.quickshift_core(sm * ratio, noise, ks, md)
        assert np.array_equal(out, z[f"labels{i}"]), f"case {i}: {(out != z[f'labels{i}']).sum()} px differ"


@pytest.mark.parametrize("name", ["c2s_256x256x4_c10", "c3s_384x384x8_c025", "ragged_200x333x5"])
def test_zonal_stats_checker_matches_numpy_golden(oracle, name):
    z, params = load(name)
    raw = z["raw"].astype(np.float32)
    st = oracle.zonal_stats_numpy(raw, z["labels"], start_label=params.get("start_label", 1))
    assert np.array_equal(st["count"], z["z_count"])
    for k, g in (("mean", "z_mean"), ("variance", "z_var"), ("min", "z_min"), ("max", "z_max")):
        np.testing.assert_allclose(st[k], z[g], rtol=1e-6, atol=0)
    # the float64 C port agrees with the float32 NumPy statistics to the north_star tolerance (1e-5 rel)
    sc = oracle.zonal_stats_c(raw, z["labels"].astype(np.int64), start_label=params.get("start_label", 1))
    assert np.array_equal(sc["count"], z["z_count"])
    np.testing.assert_allclose(sc["mean"], z["z_mean"], rtol=1e-5)
    np.testing.assert_allclose(sc["variance"], z["z_var"], rtol=1e-4, atol=1e-6 * 65535)
    np.testing.assert_array_equal(sc["min"], z["z_min"])
    np.testing.assert_array_equal(sc["max"], z["z_max"])


def test_oracle_skew_kurtosis_match_scipy_golden():
    """oracle.skew_kurtosis restates scipy.stats.skew / kurtosis (segment_statistics.py:173-175); the fixture was
    produced by SciPy 1.15.3 on the float32 pixels of every label (tests/golden/gen_goldens_moments.py)."""
    import os
    from oracle import oracle
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "moments_96x131x5.npz"))
    raw = z["dn"].astype(np.float32)
    nanmask = np.unpackbits(z["nanmask"])[:raw.size].reshape(raw.shape).astype(bool)
    raw[nanmask] = np.nan
    st = oracle.zonal_stats_numpy(raw, z["labels"])
    assert np.array_equal(np.isnan(st["skewness"]), np.isnan(z["skewness"]))
    assert np.array_equal(np.isnan(st["kurtosis"]), np.isnan(z["kurtosis"]))
    np.testing.assert_allclose(st["skewness"], z["skewness"], rtol=1e-6, atol=1e-6, equal_nan=True)
    np.testing.assert_allclose(st["kurtosis"], z["kurtosis"], rtol=1e-6, atol=1e-6, equal_nan=True)


def test_oracle_glcm_matches_skimage_golden():
    """oracle/glcm.py restates greycomatrix / greycoprops (segment_statistics.py:260-296) on the quantised bounding-box
    crop of every segment; the fixture was produced by scikit-image 0.18.3 (tests/golden/gen_goldens_glcm.py)."""
    import os
    from oracle import glcm
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "glcm_90x110x3.npz"))
    raw = z["dn"].astype(np.float32)
    nanmask = np.unpackbits(z["nanmask"])[:raw.size].reshape(raw.shape).astype(bool)
    raw[nanmask] = np.nan
    st = glcm.texture_stats(raw, z["labels"])
    for p in glcm.PROPS:
        assert np.array_equal(np.isnan(st[p]), np.isnan(z[p])), p
        np.testing.assert_allclose(st[p], z[p], rtol=1e-10, atol=1e-12, equal_nan=True, err_msg=p)
