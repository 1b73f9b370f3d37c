"""Known-answer tests of the CPU oracle on the block images scikit-image's own suite pins
(skimage/segmentation/tests/test_slic.py; see tests/scenarios.py)."""
import numpy as np
import pytest

from tests import scenarios as sc


@pytest.mark.parametrize("case", sc.slic_scenarios(), ids=lambda c: c[0])
def test_block_images(oracle, case):
    name, img, kw, expected, n_unique = case
    seg = oracle.slic(img, **kw)
    assert seg.shape == img.shape[:2]
    sc.check_expected(seg, expected, n_unique)


def test_enforce_connectivity_small(oracle):
    kw = dict(n_segments=2, compactness=0.0001, convert2lab=False, start_label=0)
    assert np.array_equal(oracle.slic(sc.CONNECTIVITY_IMG, enforce_connectivity=True, **kw), sc.CONNECTIVITY_CONNECTED)
    # float32 (obia's dtype): at compactness 1e-4 the colour term is ~1e8 and swallows the spatial term of
    # the first sweep, so the two labels come out swapped relative to scikit-image's float64 test;
    # scikit-image 0.18.3 on float32 input gives exactly this (checked in the build container).
    dis = oracle.slic(sc.CONNECTIVITY_IMG, enforce_connectivity=False, **kw)
    assert np.array_equal(dis, 1 - sc.CONNECTIVITY_DISCONNECTED)
    assert np.array_equal(oracle.slic(sc.CONNECTIVITY_IMG, enforce_connectivity=True, max_size_factor=0.8, **kw),
                          sc.CONNECTIVITY_CONNECTED)


def test_more_segments_than_pixels(oracle):
    img = sc.gray_blocks()
    seg = oracle.slic(img, n_segments=500, compactness=1, convert2lab=False, start_label=0)
    assert np.all(seg.ravel() == np.arange(seg.size))


def test_regular_grid_matches_documented_examples(oracle):
    # util/_regular_grid.py docstring examples, restated for (1,H,W): 2-D grid behaviour
    assert oracle.regular_grid(20, 40, 8) == (5, 10, 5, 10)
    assert oracle.regular_grid(512, 512, 500)[1] == 23 and oracle.regular_grid(512, 512, 500)[0] == 11
    assert oracle.regular_grid(4096, 4096, 50000)[1] == 18
    assert oracle.regular_grid(2, 3, 100) == (0, 0, 0, 0)


def test_masked_seed_rule_properties(oracle):
    """The build's deterministic maskSLIC seeding rule: seeds lie on valid pixels, count is
    close to n_segments for a filled mask, an all-ones mask reproduces the plain grid."""
    m = np.ones((96, 96), np.uint8)
    yx, steps = oracle.masked_grid_centroids(m, 30)
    g, gs = oracle.grid_centroids(96, 96, 30)
    assert np.array_equal(yx, g) and np.array_equal(steps, gs)
    yy, xx = np.mgrid[0:128, 0:160]
    m = ((yy - 60) ** 2 + (xx - 80) ** 2 < 55 ** 2).astype(np.uint8)
    yx, steps = oracle.masked_grid_centroids(m, 60)
    assert all(m[y, x] for y, x in yx)
    assert 45 <= len(yx) <= 75
    m[:] = 0
    m[5, 7] = 1
    yx, _ = oracle.masked_grid_centroids(m, 10)
    assert yx.tolist() == [[5, 7]]
