"""slic(..., sigma=...) on the GPU: the Gaussian pre-smoothing of scikit-image (scipy.ndimage.gaussian_filter between the Lab conversion
and the scaling; slic_superpixels.py) against scikit-image's own output (tests/golden/sigma*.npz, tests/golden/gen_goldens_sigma.py) and,
through the tiled driver, against the oracle's tiler (every tile smooths its own window, reflecting at the window's edges)."""
import ast
import glob
import os

import numpy as np
import pytest

from tests.metrics import adjusted_rand_index, label_disagreement

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "sigma*.npz")) + glob.glob(os.path.join(GOLD, "spacing*.npz")))


@pytest.mark.parametrize("name", CASES)
def test_slic_with_sigma_vs_skimage_golden(name):
    from obia_amd.segmentation import slic
    z = np.load(os.path.join(GOLD, name + ".npz"))
    params = ast.literal_eval(str(z["params"]))
    raw = torch.as_tensor(z["raw"].astype(np.float32)).cuda()
    sig = float(z["sigma_arg"]) if z["sigma_arg"].shape == () else [float(v) for v in z["sigma_arg"]]   # as the caller wrote it
    sp = [float(v) for v in z["spacing_zyx"]]
    kw = dict(n_segments=params["n_segments"], compactness=params["compactness"], sigma=sig, _normalize_bands=True)
    if sp != [1.0, 1.0, 1.0]:
        kw["spacing"] = sp
    if "mask" in z.files:
        kw.update(mask=z["mask"], seeds=(z["seeds_yx"], z["seed_steps_all"]))
    pre = slic(raw, enforce_connectivity=False, _stage="pre", **kw).cpu().numpy()
    lab = slic(raw, **kw).cpu().numpy()
    lab_case = z["raw"].shape[2] == 3
    if lab_case:      # float32 Lab: the tolerance of the unsmoothed 3-band cases
        assert label_disagreement(pre, z["labels_pre"]) <= 5e-4
        assert adjusted_rand_index(lab, z["labels"]) >= 0.99
    else:
        assert np.array_equal(pre, z["labels_pre"]), f"{(pre != z['labels_pre']).sum()} px differ before connectivity"
        assert np.array_equal(lab, z["labels"]), f"{(lab != z['labels']).sum()} px differ"


def test_scalar_sigma_is_the_same_width_on_every_axis(oracle):
    """sigma=1.5 == sigma=[1.5, 1.5, 1.5] (the one-plane depth axis is filtered too, as scikit-image does), and both equal the oracle."""
    from obia_amd.segmentation import slic
    rs = np.random.RandomState(2)
    yy, xx = np.mgrid[0:90, 0:130].astype(np.float32)
    img = np.stack([np.sin(xx / (6 + c)) * np.cos(yy / (5 + c)) + 0.3 * rs.randn(90, 130) for c in range(5)], -1).astype(np.float32)
    a = slic(torch.as_tensor(img).cuda(), n_segments=70, compactness=0.4, sigma=1.5, _normalize_bands=True).cpu().numpy()
    b = slic(torch.as_tensor(img).cuda(), n_segments=70, compactness=0.4, sigma=[1.5, 1.5, 1.5], _normalize_bands=True).cpu().numpy()
    ref = oracle.slic(oracle.normalize(img), n_segments=70, compactness=0.4, sigma=1.5)
    assert np.array_equal(a, b) and np.array_equal(a, ref)


def test_negative_sigma_and_bad_spacing_are_refused():
    from obia_amd.segmentation import slic
    img = torch.rand((32, 32, 4), device="cuda")
    with pytest.raises(ValueError):
        slic(img, n_segments=10, sigma=-1.0)
    with pytest.raises(ValueError):
        slic(img, n_segments=10, spacing=[1, 0, 1])
    with pytest.raises(ValueError):
        slic(img, n_segments=10, spacing=[1, 2, 3, 4])
    with pytest.raises(ValueError):
        slic(img, n_segments=10, sigma=[1, 2, 3, 4])


def test_two_element_spacing_and_sigma_are_the_row_column_form():
    """scikit-image >= 0.19 (the reference pins >= 0.23.2) takes (row, column) sequences for a 2-D image: spacing gets 1 and
    sigma 0 on the one-plane depth axis.  Same labels as the three-element form of 0.18.3."""
    from obia_amd.segmentation import slic
    rs = np.random.RandomState(5)
    yy, xx = np.mgrid[0:120, 0:150].astype(np.float32)
    img = torch.as_tensor(np.stack([np.sin(xx / (6 + c)) * np.cos(yy / (5 + c)) + 0.3 * rs.randn(120, 150) for c in range(4)], -1).astype(np.float32)).cuda()
    kw = dict(n_segments=90, compactness=0.4, _normalize_bands=True)
    assert torch.equal(slic(img, spacing=(0.5, 1.75), **kw), slic(img, spacing=(1, 0.5, 1.75), **kw))
    assert torch.equal(slic(img, sigma=(2.0, 0.6), **kw), slic(img, sigma=(0.0, 2.0, 0.6), **kw))
    assert torch.equal(slic(img, sigma=(1.0, 1.0), spacing=(2.0, 0.7), **kw), slic(img, sigma=(0, 1.0, 1.0), spacing=(1, 2.0, 0.7), **kw))


def test_unit_spacing_is_the_default_path(oracle):
    """spacing=(1, 1, 1) must not change a label (it is the default), and an anisotropic one must equal the oracle -- through the tiled
    driver as well (every tile forwards it)."""
    from obia_amd.segmentation import slic
    from obia_amd.tiling import create_tiled_segments
    from oracle import tiler
    rs = np.random.RandomState(4)
    H, W = 200, 260
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    img = np.stack([np.sin(xx / (6 + c)) * np.cos(yy / (5 + c)) + 0.3 * rs.randn(H, W) for c in range(4)], -1).astype(np.float32)
    a = slic(torch.as_tensor(img).cuda(), n_segments=150, compactness=0.4, _normalize_bands=True).cpu().numpy()
    b = slic(torch.as_tensor(img).cuda(), n_segments=150, compactness=0.4, spacing=[1, 1, 1], _normalize_bands=True).cpu().numpy()
    assert np.array_equal(a, b)
    c = slic(torch.as_tensor(img).cuda(), n_segments=150, compactness=0.4, spacing=[1, 0.5, 1.75], sigma=0.8, _normalize_bands=True).cpu().numpy()
    ref = oracle.slic(oracle.normalize(img), n_segments=150, compactness=0.4, spacing=[1, 0.5, 1.75], sigma=0.8)
    assert np.array_equal(c, ref) and not np.array_equal(c, a)
    kw = dict(tile_size=96, buffer=12, crown_radius=4, pixel_size=(0.5, 0.5), compactness=0.5, spacing=[1.0, 2.0, 0.7])
    tref, n_ref = tiler.create_tiled_segments(img, None, **kw)
    tl, n = create_tiled_segments(torch.as_tensor(img).cuda(), **kw)
    assert n == n_ref and np.array_equal(tl.cpu().numpy(), tref)


@pytest.mark.parametrize("sigma", [1.0, [0.0, 2.0, 0.6]])
def test_tiled_driver_with_sigma_vs_oracle_tiler(sigma):
    from obia_amd.tiling import create_tiled_segments
    from oracle import tiler
    rs = np.random.RandomState(7)
    H, W, C = 300, 340, 4
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    img = np.stack([200 * np.sin(xx / (9 + c)) * np.cos(yy / (7 + c)) + 500 + rs.normal(0, 30, (H, W)) for c in range(C)], -1).astype(np.float32)
    mask = np.ones((H, W), np.uint8)
    mask[120:150, 40:200] = 0
    kw = dict(tile_size=128, buffer=16, crown_radius=4, pixel_size=(0.5, 0.5), compactness=0.5)
    ref, n_ref = tiler.create_tiled_segments(img, mask, sigma=sigma, **kw)
    lab, n = create_tiled_segments(torch.as_tensor(img).cuda(), input_mask=torch.as_tensor(mask).cuda(), sigma=sigma, **kw)
    lab = lab.cpu().numpy()
    assert n == n_ref and np.array_equal(lab, ref), f"{(lab != ref).sum()} px differ"
