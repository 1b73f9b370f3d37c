"""The centroid step between two sweeps exists twice: `slic_prep_lane_kernel` (one lane per centroid, the default since round 3) and
`slic_prep_kernel` (16 / 32 lanes per centroid; OBIA_PREP_GROUPED=1).  Same arithmetic: the labels must not depend on the choice --
on batches of many problems with thousands of centroids each (where the first version of the lane kernel went wrong at random, see
tools/check_shift64.py) and for every channel-count class of the record layout."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def both(img, **kw):
    from obia_amd.tiling import create_tiled_segments
    old = os.environ.pop("OBIA_PREP_GROUPED", None)
    try:
        lab, n = create_tiled_segments(img, **kw)
        os.environ["OBIA_PREP_GROUPED"] = "1"
        ref, n_ref = create_tiled_segments(img, **kw)
    finally:
        os.environ.pop("OBIA_PREP_GROUPED", None)
        if old is not None:
            os.environ["OBIA_PREP_GROUPED"] = old
    return lab, n, ref, n_ref


@pytest.mark.parametrize("bands", [3, 8, 9, 13])
def test_lane_and_grouped_prep_kernels_agree_on_a_tiled_raster(bands):
    rs = np.random.RandomState(bands)
    H, W = 900, 1100
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    img = np.stack([300 * np.sin(xx / (8 + c)) * np.cos(yy / (6 + c)) + 900 + rs.normal(0, 25, (H, W)) for c in range(bands)], -1).astype(np.float32)
    mask = np.ones((H, W), np.uint8)
    mask[400:430, 200:700] = 0
    lab, n, ref, n_ref = both(torch.as_tensor(img).cuda(), input_mask=torch.as_tensor(mask).cuda(), tile_size=256, buffer=32, crown_radius=4,
                              pixel_size=(0.5, 0.5), compactness=10.0)
    assert n == n_ref and torch.equal(lab, ref)


def test_lane_and_grouped_prep_kernels_agree_on_32_problems_of_13000_centroids():
    """8192 x 8192 x 8 at the bench's tiling: the black batch holds 8 problems of 12 996 centroids, the white rows 4 of ~14 000 --
    three repetitions, the failure this guards against hit ~1 % of the centroids, different ones every time."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import synth_raster
    dev = torch.device("cuda:0")
    img = synth_raster(8192, 8192, 8, seed=3, device=dev)
    mask = torch.ones((8192, 8192), dtype=torch.uint8, device=dev)
    for _ in range(3):
        lab, n, ref, n_ref = both(img, input_mask=mask, tile_size=2048, buffer=64, crown_radius=5, pixel_size=(0.5, 0.5), compactness=10.0)
        assert n == n_ref and torch.equal(lab, ref)
