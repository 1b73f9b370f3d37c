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


def test_sweep_timing_sum_and_union():
    """obia_last_timing classes 0 / 10: the colour sweeps' durations (events bound to the dispatches) add up to the time during
    which one of them runs when they run one after the other; with two groups side by side the union is shorter than the sum."""
    from obia_amd import _lib
    from obia_amd.tiling import create_tiled_segments
    rs = np.random.RandomState(5)
    img = torch.as_tensor(rs.rand(1024, 1280, 8).astype(np.float32)).cuda()
    ctx = _lib.Context(0)
    ctx.set_profiling(2)
    kw = dict(tile_size=256, buffer=32, crown_radius=4, pixel_size=(0.5, 0.5), compactness=10.0, ctx=ctx)
    old = os.environ.pop("OBIA_SWEEP_GROUPS", None)
    try:
        create_tiled_segments(img, **kw)
        t1 = ctx.timing()
        os.environ["OBIA_SWEEP_GROUPS"] = "2"
        create_tiled_segments(img, **kw)
        t2 = ctx.timing()
    finally:
        os.environ.pop("OBIA_SWEEP_GROUPS", None)
        if old is not None:
            os.environ["OBIA_SWEEP_GROUPS"] = old
    assert t1["sweeps"] > 0 and t1["assign_ms"] > 0
    assert abs(t1["assign_busy_ms"] - t1["assign_ms"]) <= 0.01 * t1["assign_ms"] + 0.005
    assert t2["sweeps"] == 2 * t1["sweeps"]                     # every batch of this raster holds at least two tiles
    assert t2["assign_px"] == t1["assign_px"]                   # the same pixels, dealt into two launches per sweep
    assert t2["assign_busy_ms"] <= t2["assign_ms"] * 1.001 + 0.005
