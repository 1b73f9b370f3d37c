"""OBIA_SWEEP_GROUPS (slic_run_sweeps): the problems of a batch dealt into 2..4 groups whose prep / sweep chains run on streams of
their own.  The grouping must not change a label: same kernels, same per-problem integer accumulators."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("groups", [2, 3, 4])
@pytest.mark.parametrize("compactness", [10.0, 0.25])
def test_grouped_sweeps_give_identical_tiled_labels(groups, compactness):
    from obia_amd.tiling import create_tiled_segments
    rs = np.random.RandomState(77)
    H, W, C = 700, 900, 8
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    img = np.stack([200 * np.sin(xx / (9 + c)) * np.cos(yy / (7 + c)) + 500 + rs.normal(0, 20, (H, W)) for c in range(C)], -1).astype(np.float32)
    mask = np.ones((H, W), np.uint8)
    mask[300:340, 100:500] = 0
    kw = dict(input_mask=torch.as_tensor(mask).cuda(), tile_size=192, buffer=24, crown_radius=4, pixel_size=(0.5, 0.5), compactness=compactness)
    dev = torch.as_tensor(img).cuda()
    old = os.environ.pop("OBIA_SWEEP_GROUPS", None)
    try:
        ref, n_ref = create_tiled_segments(dev, **kw)
        os.environ["OBIA_SWEEP_GROUPS"] = str(groups)
        lab, n = create_tiled_segments(dev, **kw)
    finally:
        os.environ.pop("OBIA_SWEEP_GROUPS", None)
        if old is not None:
            os.environ["OBIA_SWEEP_GROUPS"] = old
    assert n == n_ref and torch.equal(lab, ref)
