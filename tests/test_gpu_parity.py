"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI
(libobia_hip.so via obia_amd), against the CPU oracle and the committed scikit-image golden vectors.

Stated tolerances (north_star: "label maps within a stated adjusted-Rand / boundary-recall tolerance,
per-segment band statistics within 1e-5 relative"):
  * labels vs the scikit-image goldens: BIT-EXACT before and after connectivity for every case that is not on the
    explicit allow-list NOT_BIT_EXACT below (one Lab case, a handful of pixels);
  * pre-connectivity labels vs the ORACLE on random inputs: <= 1e-4 of pixels differ on non-Lab inputs (the only
    source of difference is the rounding of centroid colour means: the reference accumulates them
    sequentially in float32, the HIP path in exact 64-bit fixed point), <= 5e-4 on 3-band Lab inputs
    (device powf/cbrtf vs libm);
  * final labels: ARI >= 0.99, boundary recall and precision (1 px) >= 0.99, segment count within 1 %;
  * connectivity enforcement given identical input labels: bit-exact (no component reaches max_size);
  * zonal statistics given identical labels: 1e-5 relative (variance: + 1e-6 * range^2 absolute).
"""
import ast
import glob
import os

import numpy as np
import pytest

from tests.metrics import (adjusted_rand_index, boundary_recall_precision, check_connected_consecutive,
                           label_disagreement)
from tests import scenarios as sc

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SLIC_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "*.npz"))
                    if not os.path.basename(p).startswith(("connectivity_", "quickshift_", "moments_", "glcm_", "sigma", "spacing")))
UNMASKED = [c for c in SLIC_CASES if not c.startswith("mask")]


@pytest.fixture(scope="module")
def amd():
    import obia_amd
    from obia_amd import _lib
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    _lib.load()
    return obia_amd


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    return z, ast.literal_eval(str(z["params"]))


def kwargs_of(params):
    kw = dict(n_segments=params["n_segments"], compactness=params["compactness"],
              max_num_iter=params.get("max_iter", 10), convert2lab=params.get("convert2lab", None),
              min_size_factor=params.get("min_size_factor", 0.5), max_size_factor=params.get("max_size_factor", 3),
              start_label=params.get("start_label", 1))
    if params.get("slic_zero"):
        kw["slic_zero"] = True
    return kw


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


# Every scikit-image golden of the grid path is reproduced BIT FOR BIT, before and after connectivity -- with the exceptions listed
# here, each with the largest number of differing pixels accepted (before, after connectivity).  Measured by
# tools/golden_exactness.py (round 4: c1_512x512x3 6 / 3 px of 262 144; every other case 0 / 0).  A case that is not listed must
# be array_equal: a regression from 0 differing pixels to "within the stated tolerance" fails (Weak 2).
NOT_BIT_EXACT = {
    # sRGB -> Lab on the device: powf / cbrtf of the HIP runtime against the libm scikit-image was built on (last-bit differences in
    # a few features move a handful of near-tie pixels).  The other Lab golden, quickstart_128x128x3, is exact.
    "c1_512x512x3": (8, 5),
}


@pytest.mark.parametrize("name", UNMASKED)
def test_slic_pre_connectivity_vs_golden(amd, name):
    from obia_amd.segmentation import slic
    z, params = load(name)
    raw = dev(z["raw"].astype(np.float32))
    pre = slic(raw, enforce_connectivity=False, _normalize_bands=True, _stage="pre", **{k: v for k, v in kwargs_of(params).items()
                                                                                       if k not in ("min_size_factor", "max_size_factor")})
    pre = pre.cpu().numpy()
    nd = int((pre != z["labels_pre"]).sum())
    if name in NOT_BIT_EXACT:
        assert nd <= NOT_BIT_EXACT[name][0], f"{name}: {nd} pixels differ before connectivity (allow-list: {NOT_BIT_EXACT[name][0]})"
        assert label_disagreement(pre, z["labels_pre"]) <= 5e-4    # the stated tolerance of the Lab cases
    else:
        assert nd == 0, f"{name}: {nd} pixels differ before connectivity (the case is not on the allow-list: it must be bit-exact)"


@pytest.mark.parametrize("name", UNMASKED)
def test_slic_final_vs_golden(amd, name):
    from obia_amd.segmentation import slic
    z, params = load(name)
    raw = z["raw"].astype(np.float32)
    # host-pointer entry (NumPy in / NumPy out), as the reference's caller would use it
    lab = slic(raw, _normalize_bands=True, **kwargs_of(params))
    gold = z["labels"]
    assert lab.dtype == np.int64 and lab.shape == gold.shape
    # bit-exact unless the case is on the allow-list above ("sizefac": max_size_factor 1.2 makes the reference cut most components
    # at max_size -- capped BFS, cc_split_kernel -- and the cut pieces are exact too)
    nd = int((lab != gold).sum())
    if name in NOT_BIT_EXACT:
        assert nd <= NOT_BIT_EXACT[name][1], f"{name}: {nd} pixels differ after connectivity (allow-list: {NOT_BIT_EXACT[name][1]})"
    else:
        assert nd == 0, f"{name}: {nd} pixels differ after connectivity (the case is not on the allow-list: it must be bit-exact)"
    ari = adjusted_rand_index(lab, gold)
    rec, prec = boundary_recall_precision(gold, lab)
    n_g, n_l = len(np.unique(gold)), len(np.unique(lab))
    assert ari >= 0.99, f"{name}: ARI {ari}"
    assert rec >= 0.99 and prec >= 0.99, f"{name}: boundary recall {rec} precision {prec}"
    assert abs(n_g - n_l) <= max(1, 0.01 * n_g)
    if "morethanpx" not in name:
        check_connected_consecutive(lab, params.get("start_label", 1))


def test_slic_matches_oracle_bit_exact_when_centroids_agree(amd, oracle):
    """compactness 10 on 4 bands: the spatial term dominates, centroid rounding cannot flip a pixel --
    the HIP labels must equal the oracle's exactly, before and after connectivity."""
    from obia_amd.segmentation import slic
    z, params = load("c2s_256x256x4_c10")
    raw = z["raw"].astype(np.float32)
    lab = slic(raw, _normalize_bands=True, **kwargs_of(params))
    o_lab, o_pre, _ = oracle.slic(oracle.normalize(raw), n_segments=params["n_segments"], compactness=params["compactness"],
                                  return_all=True)
    assert np.array_equal(lab, o_lab)
    pre = slic(dev(raw), _normalize_bands=True, _stage="pre", n_segments=params["n_segments"],
               compactness=params["compactness"]).cpu().numpy()
    assert np.array_equal(pre, o_pre)


@pytest.mark.parametrize("name", ["c2s_256x256x4_c10", "c3s_384x384x8_c025", "c2s_256x256x4_c005", "mask_128x160x4_c10",
                                  "quickstart_128x128x3", "iter3_100x120x4"])
def test_exit_on_fixed_point_is_bit_identical(amd, name):
    """Stopping once the centroid records repeat must give exactly the labels of all max_num_iter sweeps."""
    from obia_amd.segmentation import slic
    from obia_amd.tiling import create_tiled_segments
    z, params = load(name)
    raw = z["raw"].astype(np.float32)
    mask = z["mask"] if "mask" in z.files else None
    a = slic(raw, mask=mask, _normalize_bands=True, **kwargs_of(params))
    b = slic(raw, mask=mask, _normalize_bands=True, exit_on_fixed_point=True, **kwargs_of(params))
    assert np.array_equal(a, b)
    kw = dict(tile_size=100, buffer=16, crown_radius=4, pixel_size=(1.0, 1.0), compactness=params["compactness"])
    la, na = create_tiled_segments(raw, input_mask=mask, **kw)
    lb, nb = create_tiled_segments(raw, input_mask=mask, exit_on_fixed_point=True, **kw)
    assert na == nb and np.array_equal(la, lb)


@pytest.mark.parametrize("shape,n_seg,comp", [((96, 96, 4), 576, 10.0), ((130, 70, 8), 1200, 0.5), ((64, 200, 3), 3000, 1.0)])
def test_dense_seeds_take_the_direct_bin_path(amd, oracle, shape, n_seg, comp):
    """S = 2..4 pixels: a 64x64 tile meets hundreds of candidate centroids, more than the LDS slots of the sweep
    kernel, so the tile takes slow_tile() (bins read directly from global memory).  Same arithmetic: the labels must
    still equal the oracle's."""
    from obia_amd.segmentation import slic
    rs = np.random.RandomState(11)
    H, W, C = shape
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    raw = np.stack([400 * np.sin(xx / (5 + c)) * np.cos(yy / (7 + c)) + 1000 + rs.normal(0, 30, (H, W)) for c in range(C)],
                   -1).astype(np.float32)
    kw = dict(n_segments=n_seg, compactness=comp, convert2lab=False)
    lab = slic(raw, _normalize_bands=True, **kw)
    ref, pre, _ = oracle.slic(oracle.normalize(raw), return_all=True, **kw)
    got_pre = slic(dev(raw), _normalize_bands=True, _stage="pre", **kw).cpu().numpy()
    assert label_disagreement(got_pre, pre) <= 1e-4
    assert adjusted_rand_index(lab, ref) >= 0.99


@pytest.mark.parametrize("name", [c for c in SLIC_CASES if c.startswith("mask")])
def test_masked_slic_vs_oracle_same_rule(amd, oracle, name):
    """maskSLIC: the HIP path and the oracle use the same deterministic masked-grid seeding,
    so they are compared with each other; the scikit-image golden (RNG + kmeans2 seeding) is compared
    statistically (segment count within 25 %, every masked pixel labelled 0)."""
    from obia_amd.segmentation import slic
    z, params = load(name)
    raw = z["raw"].astype(np.float32)
    mask = z["mask"]
    lab = slic(raw, mask=mask, _normalize_bands=True, **kwargs_of(params))
    o_lab = oracle.slic(oracle.normalize(raw), mask=mask, n_segments=params["n_segments"],
                        compactness=params["compactness"])
    assert (lab[mask == 0] == 0).all() and (lab[mask != 0] > 0).all()
    ari = adjusted_rand_index(lab, o_lab)
    assert ari >= 0.99, f"{name}: ARI vs oracle {ari}"
    n_gold = len(np.unique(z["labels"])) - 1
    n_lab = len(np.unique(lab)) - 1
    assert abs(n_lab - n_gold) <= 0.25 * n_gold


@pytest.mark.parametrize("name", [c for c in SLIC_CASES if c.startswith("mask")])
def test_masked_slic_on_skimage_seeds_vs_golden(amd, name):
    """maskSLIC pinned on scikit-image itself: the HIP path is fed scikit-image's OWN seeds (the `seeds_yx` /
    `seed_steps_all` the golden generator took from `_get_mask_centroids`, tests/golden/gen_goldens.py) through the
    seeds input of the C ABI (obia_slic_seeded_f32_dev) and compared with scikit-image's OUTPUT, not with the oracle:
    spatial-only pre-pass + main pass (slic_superpixels.py:310-318) -> labels identical before and after connectivity.
    This is the path every tile of create_tiled_segments takes (tiling.py:121-143)."""
    from obia_amd.segmentation import slic
    z, params = load(name)
    raw = dev(z["raw"].astype(np.float32))
    mask = z["mask"]
    seeds = (z["seeds_yx"], z["seed_steps_all"])
    kw = kwargs_of(params)
    pre = slic(raw, mask=mask, seeds=seeds, _normalize_bands=True, _stage="pre", **kw).cpu().numpy()
    lab = slic(raw, mask=mask, seeds=seeds, _normalize_bands=True, **kw).cpu().numpy()
    gold = z["labels"]
    assert (lab[mask == 0] == 0).all() and (lab[mask != 0] > 0).all()
    # bit-exact before and after connectivity on all three masked goldens (tools/golden_exactness.py, round 4: 0 / 0 pixels); a case
    # that stops being exact has to be put on the allow-list with its count, not waved through a tolerance
    n_pre, n_fin = int((pre != z["labels_pre"]).sum()), int((lab != gold).sum())
    a_pre, a_fin = NOT_BIT_EXACT.get(name, (0, 0))
    assert n_pre <= a_pre, f"{name}: {n_pre} pixels differ before connectivity (scikit-image seeds)"
    assert n_fin <= a_fin, f"{name}: {n_fin} pixels differ after connectivity (scikit-image seeds)"


def test_seeded_slic_rejects_bad_seeds(amd):
    from obia_amd.segmentation import slic
    img = dev(np.random.RandomState(0).rand(32, 40, 4).astype(np.float32))
    with pytest.raises(ValueError):
        slic(img, seeds=(np.array([[5.0, 50.0]]), [1.0, 8.0, 8.0]), _normalize_bands=True)    # x outside the raster
    with pytest.raises(ValueError):
        slic(img, seeds=(np.zeros((0, 2)), [1.0, 8.0, 8.0]), _normalize_bands=True)
    with pytest.raises(ValueError):
        slic(img.cpu().numpy(), seeds=(np.array([[5.0, 5.0]]), [1.0, 8.0, 8.0]), _normalize_bands=True)   # host arrays: no seeds


def test_seeded_slic_on_an_empty_mask_is_refused_before_any_upload(amd):
    """caller-supplied seeds on a mask without a valid pixel used to upload n seeds into a one-record array
    (device-side overflow) before OBIA_E_EMPTY was returned.  Refused up front now; the context stays usable."""
    from obia_amd.segmentation import slic
    img = dev(np.random.RandomState(0).rand(48, 56, 4).astype(np.float32))
    seeds = (np.stack([np.linspace(2, 45, 300), np.linspace(2, 53, 300)], 1), [1.0, 6.0, 6.0])
    for _ in range(3):
        with pytest.raises(ValueError):
            slic(img, mask=np.zeros((48, 56), bool), seeds=seeds, _normalize_bands=True)
    ok = slic(img, mask=np.ones((48, 56), bool), seeds=seeds, _normalize_bands=True)      # same context, same seeds
    assert int(ok.max().item()) >= 1


def test_orphan_that_first_appears_in_the_very_last_sweep(amd, oracle):
    """only the last sweep of a batch stores labels; a valid pixel that no window reaches ("orphan") keeps the
    label of the sweep before, so the sweeps are repeated with every sweep storing.  The flag that triggers the repeat was
    not raised by the LAST sweep itself: a pixel orphaned there for the first time came out with the fill value.
    Case built with the oracle (tools: per-sweep coverage): a two-pixel island 2 * step rows above a ragged-edged region,
    max_num_iter = 2 -- covered in both pre-pass sweeps and in the first colour sweep, out of every window in the second."""
    from obia_amd.segmentation import slic
    rs = np.random.RandomState(114)
    H, W, C = int(rs.choice([80, 96, 110])), int(rs.choice([100, 120, 140])), int(rs.choice([1, 2, 4]))
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    img = np.stack([300 * np.sin(xx / (5 + 2 * c)) * np.cos(yy / (6 + c)) + 800 + 40 * c + rs.normal(0, 25, (H, W))
                    for c in range(C)], -1).astype(np.float32)
    n_seg = int(rs.choice([40, 60, 90])); top = int(rs.choice([40, 44, 48, 52])); comp = float(rs.choice([5.0, 20.0]))
    mask = np.zeros((H, W), np.uint8); mask[top:, :] = 1
    for x in range(W):
        mask[top - int(rs.randint(0, 6)):top, x] = 1
    assert (H, W, C, n_seg, top, comp) == (110, 140, 4, 60, 44, 5.0)
    mask[11, 20:22] = 1                                   # the island
    kw = dict(n_segments=n_seg, compactness=comp, convert2lab=False, start_label=1)
    ref, ref_pre, _ = oracle.slic(oracle.normalize(img), return_all=True, max_iter=2, mask=mask, **kw)
    one, one_pre, _ = oracle.slic(oracle.normalize(img), return_all=True, max_iter=1, mask=mask, **kw)
    assert (ref_pre[11, 20:22] >= 1).all(), "the oracle keeps the label of the sweep before for the island"
    got_pre = slic(dev(img), mask=mask.astype(bool), _normalize_bands=True, _stage="pre", max_num_iter=2, **kw).cpu().numpy()
    assert np.array_equal(got_pre[11, 20:22], ref_pre[11, 20:22])
    assert np.array_equal(got_pre, ref_pre)
    got = slic(dev(img), mask=mask.astype(bool), _normalize_bands=True, max_num_iter=2, **kw).cpu().numpy()
    assert np.array_equal(got, ref)
    # with the fixed-point cache and with SLIC-zero every sweep stores from the start: same answer
    assert np.array_equal(slic(dev(img), mask=mask.astype(bool), _normalize_bands=True, max_num_iter=2, exit_on_fixed_point=True, **kw).cpu().numpy(), ref)


def test_seeded_grid_equals_library_seeding(amd, oracle):
    """Feeding the plain grid through the seeds input must reproduce the library's own grid seeding bit for bit."""
    from obia_amd.segmentation import slic
    z, params = load("c2s_256x256x4_c10")
    raw = dev(z["raw"].astype(np.float32))
    H, W = z["raw"].shape[:2]
    yx, steps = oracle.grid_centroids(H, W, params["n_segments"])
    a = slic(raw, _normalize_bands=True, **kwargs_of(params))
    b = slic(raw, seeds=(yx.astype(np.float64), steps), _normalize_bands=True, **kwargs_of(params))
    assert torch.equal(a, b)


def test_connectivity_stage_bit_exact_vs_oracle(amd, oracle):
    from obia_amd.segmentation import enforce_connectivity
    z = np.load(os.path.join(GOLD, "connectivity_blackbox.npz"))
    rs = np.random.RandomState(5)
    cases = [(z[f"in{i}"], int(z[f"par{i}"][0])) for i in range(5)]
    # plus fragment-rich SLIC label maps from the goldens
    for name in ("c2s_256x256x4_c005", "c3s_384x384x8_c025", "ragged_200x333x5"):
        g, _ = load(name)
        cases.append((g["labels_pre"], 40))
    lab = (rs.randint(0, 5, (97, 131)) + 1).astype(np.int32)   # salt-and-pepper: thousands of tiny components
    cases.append((lab, 4))
    for lab_in, mn in cases:
        # max_size: never reached / the reference's default ratio to min_size (6) / cuts most components / cuts nearly
        # everything / below min_size (every piece is "small") / single pixels
        for mx in (lab_in.size + 1, 6 * mn, 3 * mn, mn + 3, max(1, mn // 2), 1):
            ref = oracle.enforce_connectivity(lab_in.astype(np.int64), mn, mx, start_label=1)
            out, n = enforce_connectivity(dev(lab_in.astype(np.int32)), mn, mx, start_label=1)
            out = out.cpu().numpy()
            assert np.array_equal(out, ref), f"{(out != ref).sum()} px differ (min_size {mn}, max_size {mx}, shape {lab_in.shape})"
            assert n == len(np.unique(ref[ref > 0]))


@pytest.mark.parametrize("name", ["c2s_256x256x4_c10", "c3s_384x384x8_c025", "ragged_200x333x5", "onech_90x110x1"])
def test_zonal_stats_vs_numpy_golden(amd, name):
    from obia_amd.statistics import zonal_stats
    z, params = load(name)
    raw = z["raw"].astype(np.float32)
    st = zonal_stats(raw, z["labels"], start_label=params.get("start_label", 1))
    assert np.array_equal(st["count"], z["z_count"])
    np.testing.assert_allclose(st["mean"], z["z_mean"], rtol=1e-5)
    rng = float(raw.max() - raw.min())
    np.testing.assert_allclose(st["variance"], z["z_var"], rtol=1e-5, atol=1e-6 * rng * rng)
    np.testing.assert_array_equal(st["min"], z["z_min"].astype(np.float32))
    np.testing.assert_array_equal(st["max"], z["z_max"].astype(np.float32))
    # device-tensor entry gives the same table
    st2 = zonal_stats(dev(raw), dev(z["labels"]), start_label=params.get("start_label", 1))
    np.testing.assert_allclose(st2["mean"].cpu().numpy(), st["mean"], rtol=1e-12)


def test_zonal_stats_edge_cases(amd):
    from obia_amd.statistics import zonal_stats
    raw = np.arange(6 * 7 * 3, dtype=np.float32).reshape(6, 7, 3)
    raw[0, 0, 1] = np.nan
    lab = np.zeros((6, 7), np.int32)
    lab[:3] = 1
    lab[3:] = 3                      # label 2 is empty; label 0 / -1 ignored
    lab[5, 6] = -1
    st = zonal_stats(raw, lab, bands=[1, 2], n_labels=3)
    assert st["count"].tolist() == [21, 0, 20]
    assert np.isnan(st["mean"][1]).all() and np.isnan(st["min"][1]).all()
    v = raw[:3, :, 1].ravel()
    v = v[~np.isnan(v)]
    np.testing.assert_allclose(st["mean"][0, 0], v.mean(), rtol=1e-6)
    np.testing.assert_allclose(st["variance"][0, 0], v.var(), rtol=1e-5)
    with pytest.raises(IndexError):
        zonal_stats(raw, lab, bands=[3])


@pytest.mark.parametrize("case", sc.slic_scenarios(), ids=lambda c: c[0])
def test_known_answer_block_images(amd, case):
    from obia_amd.segmentation import slic
    name, img, kw, expected, n_unique = case
    seg = slic(img, **kw)
    sc.check_expected(seg, expected, n_unique)


def test_known_answer_connectivity_and_tiny(amd):
    from obia_amd.segmentation import slic
    kw = dict(n_segments=2, compactness=0.0001, convert2lab=False, start_label=0)
    assert np.array_equal(slic(sc.CONNECTIVITY_IMG, enforce_connectivity=True, **kw), sc.CONNECTIVITY_CONNECTED)
    assert np.array_equal(slic(sc.CONNECTIVITY_IMG, enforce_connectivity=False, **kw), 1 - sc.CONNECTIVITY_DISCONNECTED)
    seg = slic(sc.gray_blocks(), n_segments=500, compactness=1, convert2lab=False, start_label=0)
    assert np.all(seg.ravel() == np.arange(seg.size))


def test_error_behaviour(amd):
    from obia_amd.segmentation import slic, create_segments
    img = np.random.RandomState(0).rand(16, 16, 4).astype(np.float32)
    with pytest.raises(ValueError):
        slic(img, start_label=2)
    with pytest.raises(ValueError):
        slic(img, convert2lab=True)           # Lab needs 3 bands
    with pytest.raises(ValueError):
        slic(img, mask=np.zeros((16, 16), np.uint8))   # empty mask
    with pytest.raises(ValueError):
        slic(img, mask=np.ones((4, 4), np.uint8))
    const = img.copy()
    const[:, :, 2] = 7.0
    with pytest.raises(ValueError):
        create_segments(const, n_segments=4)  # constant band: normalize_band would be 0/0
    with pytest.raises(IndexError):
        create_segments(img, segmentation_bands=[4], n_segments=4)
    with pytest.raises(Exception):
        create_segments(img, method="watershed")
    with pytest.raises(ValueError):
        slic(img, spacing=[1.0, -2.0, 1.0])         # (sigma and spacing are implemented since round 3: tests/test_gpu_sigma.py)
    with pytest.raises(ValueError):
        slic(img, sigma=-1.0)


def test_segment_end_to_end_quickstart(amd, oracle):
    """docs/examples/segmentation-quickstart.ipynb input through segment(): label raster + objects table."""
    from obia_amd import segment
    z, params = load("quickstart_128x128x3")

    class Img:
        img_data = z["raw"].astype(np.float32)

    before = Img.img_data.copy()
    seg = segment(Img, segmentation_bands=[0, 1, 2], method="slic", n_segments=200, compactness=8, start_label=1)
    assert np.array_equal(Img.img_data, before)        # not mutated
    lab = seg._segments
    assert adjusted_rand_index(lab, z["labels"]) >= 0.99
    tbl = seg.segments
    assert list(tbl.columns[:5]) == ["segment_id", "b0_mean", "b0_variance", "b0_min", "b0_max"]
    ref = oracle.zonal_stats_numpy(before, lab)
    np.testing.assert_allclose(tbl["b2_mean"].to_numpy(), ref["mean"][:, 2], rtol=1e-5)
    np.testing.assert_allclose(tbl["b1_variance"].to_numpy(), ref["variance"][:, 1], rtol=1e-4, atol=1e-3)


def test_zonal_skewness_kurtosis_vs_scipy_golden(amd):
    """Second zonal pass (obia_zonal_moments_f32): skewness / kurtosis per (label, band) against SciPy's own output on
    the float32 pixels (fixture from tests/golden/gen_goldens_moments.py).  SciPy computes the moments in float32, the
    kernel in float64 about the float64 mean: 1e-4 absolute / relative covers float32 rounding of m2..m4 (the data are
    a few hundred pixels per label).  NaN pattern (empty, all-NaN band, constant segment) must be identical."""
    import os
    from obia_amd.statistics import zonal_stats, create_objects
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "moments_96x131x5.npz"))
    raw = z["dn"].astype(np.float32)
    nanmask = np.unpackbits(z["nanmask"])[:raw.size].reshape(raw.shape).astype(bool)
    raw[nanmask] = np.nan
    lab = z["labels"]
    st = zonal_stats(raw, lab, moments=True)
    assert np.array_equal(np.isnan(st["skewness"]), np.isnan(z["skewness"]))
    assert np.array_equal(np.isnan(st["kurtosis"]), np.isnan(z["kurtosis"]))
    np.testing.assert_allclose(st["skewness"], z["skewness"], rtol=1e-4, atol=1e-4, equal_nan=True)
    np.testing.assert_allclose(st["kurtosis"], z["kurtosis"], rtol=1e-4, atol=1e-4, equal_nan=True)
    # device tensors and a band subset in another order give the same numbers
    st2 = zonal_stats(dev(raw), dev(lab), bands=[4, 1], moments=True)
    np.testing.assert_allclose(st2["skewness"].cpu().numpy(), st["skewness"][:, [4, 1]], rtol=1e-12, equal_nan=True)
    np.testing.assert_allclose(st2["kurtosis"].cpu().numpy(), st["kurtosis"][:, [4, 1]], rtol=1e-12, equal_nan=True)
    # the objects table carries the columns in the reference's order (segment_statistics.py:66-75)
    # ... with the reference's defaults: GLCM columns of every band, the five point-cloud columns (NaN), geometry last
    # (segment_statistics.py:96-108); one row per label that exists, segment_id 1..N (segment_boundaries.py:76)
    df = create_objects(lab, raw, spectral_bands=[0, 2], geometry=False)
    C = raw.shape[2]
    assert list(df.columns) == ["segment_id"] + [f"b{b}_{s}" for b in (0, 2)
                                                  for s in ("mean", "variance", "min", "max", "skewness", "kurtosis")] \
        + [f"b{b}_{t}" for b in range(C) for t in ("contrast", "dissimilarity", "homogeneity", "ASM", "energy", "correlation")] \
        + ["pai", "fhd", "ch", "mean_intensity", "variance_intensity", "geometry"]
    present = st["count"] > 0
    assert list(df["segment_id"]) == list(range(1, int(present.sum()) + 1))
    np.testing.assert_allclose(df["b2_skewness"].to_numpy(), st["skewness"][present, 2], rtol=1e-12, equal_nan=True)
    assert df[["pai", "fhd", "ch", "mean_intensity", "variance_intensity"]].isna().all().all()


def test_glcm_texture_vs_skimage_golden(amd, oracle):
    """obia_texture_stats_f32_dev against scikit-image's own greycomatrix / greycoprops output (fixture) -- integer pair
    sums are exact, so 1e-9 relative -- on segments that include a single pixel, a two-row strip (the vertical offset
    finds no pair), a constant segment and a band without a valid pixel; then on a coarse partition whose bounding
    boxes exceed 4096 pixels (dense-matrix path) against the CPU restatement; then through create_objects."""
    import os
    from obia_amd.statistics import texture_stats, create_objects, TEXTURE_PROPS
    from oracle import glcm
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "glcm_90x110x3.npz"))
    raw = z["dn"].astype(np.float32)
    nanmask = np.unpackbits(z["nanmask"])[:raw.size].reshape(raw.shape).astype(bool)
    raw[nanmask] = np.nan
    lab = z["labels"]
    st = texture_stats(raw, lab)
    for p in TEXTURE_PROPS:
        assert np.array_equal(np.isnan(st[p]), np.isnan(z[p])), p
        np.testing.assert_allclose(st[p], z[p], rtol=1e-9, atol=1e-12, equal_nan=True, err_msg=p)
    # big bounding boxes (> 4096 px) and a band subset, device tensors in
    yy, xx = np.mgrid[0:raw.shape[0], 0:raw.shape[1]]
    coarse = ((yy // 75) * 2 + xx // 80 + 1).astype(np.int32)
    st2 = texture_stats(dev(raw), dev(coarse), bands=[2, 0])
    chk = glcm.texture_stats(raw, coarse, bands=[2, 0])
    for p in TEXTURE_PROPS:
        np.testing.assert_allclose(st2[p].cpu().numpy(), chk[p], rtol=1e-9, atol=1e-12, equal_nan=True, err_msg=p)
    df = create_objects(lab, raw, spectral_bands=[0], textural_bands=[1], calc_skewness=False, calc_kurtosis=False,
                        calc_pai=False, calc_fhd=False, calc_ch=False, calc_mean_intensity=False, calc_variance_intensity=False)
    assert list(df.columns) == ["segment_id", "b0_mean", "b0_variance", "b0_min", "b0_max"] + [f"b1_{p}" for p in TEXTURE_PROPS] \
        + ["geometry"]
    from obia_amd.statistics import zonal_stats
    present = zonal_stats(raw, lab)["count"] > 0
    np.testing.assert_allclose(df["b1_energy"].to_numpy(), st["energy"][present, 1], rtol=1e-12, equal_nan=True)
    # calculate_textural=False keeps the columns (the reference builds them from textural_bands, :466-471) and leaves them NaN
    df2 = create_objects(lab, raw, spectral_bands=[0], textural_bands=[1], calculate_textural=False, geometry=False)
    assert df2[[f"b1_{p}" for p in TEXTURE_PROPS]].isna().all().all() and len(df2) == len(df)
