"""Randomised parity of the widened rows: skewness / kurtosis (f2), GLCM texture (f3) and the label-edge raster (f4) against their CPU
restatements (oracle.zonal_stats_numpy, oracle/glcm.py, oracle/consumers.py) on seeded rasters with NaN pixels and label maps with
unlabelled pixels, gaps in the numbering, thin and large segments (small crops take the LDS hash path of the texture kernel, large
ones the dense matrix).  Tolerances as in the fixed cases: moments 1e-3, texture 1e-9, edges exact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def make_case(seed):
    rs = np.random.RandomState(19000 + seed)
    H, W, C = int(rs.randint(4, 150)), int(rs.randint(4, 180)), int(rs.choice([1, 3, 4, 8]))
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    raw = np.stack([rs.uniform(20, 300) * np.sin(xx / (6 + c)) * np.cos(yy / (8 + c)) + rs.uniform(0, 1000) + rs.normal(0, 25, (H, W))
                    for c in range(C)], -1).astype(np.float32)
    s = int(rs.choice([3, 7, 19, 70]))      # 70: crops above 4096 pixels (dense GLCM path)
    lab = ((yy + rs.randn(H, W)) // s).astype(np.int64) * ((W + s - 1) // s + 2) + ((xx + rs.randn(H, W)) // s).astype(np.int64)
    lab = lab - lab.min() + 1
    if rs.rand() < 0.5:
        lab[lab % 4 == 0] = 0
    if rs.rand() < 0.4:
        lab[rs.rand(H, W) < 0.03] = 0
    return raw, lab.astype(np.int32)


@pytest.mark.parametrize("seed", range(int(os.environ.get("OBIA_RANDOM_WIDENED_CASES", "20"))))
def test_random_moments_texture_edges_vs_restatements(oracle, seed):
    from obia_amd.consumers import slic_edge
    from obia_amd.statistics import TEXTURE_PROPS, texture_stats, zonal_stats
    from oracle import glcm
    from oracle.consumers import edge_raster
    raw, lab = make_case(seed)
    np.testing.assert_array_equal(slic_edge(lab), edge_raster(lab).astype(np.float32))
    if lab.max() < 1:
        return
    st = zonal_stats(raw, lab, moments=True)
    chk = oracle.zonal_stats_numpy(raw, lab)
    assert np.array_equal(st["count"], chk["count"])
    for k in ("skewness", "kurtosis"):
        assert np.array_equal(np.isnan(st[k]), np.isnan(chk[k])), f"seed {seed}: NaN pattern of {k}"
        np.testing.assert_allclose(st[k], chk[k], rtol=1e-3, atol=1e-3, equal_nan=True, err_msg=f"seed {seed} {k}")
    tx = texture_stats(raw, lab)
    ref = glcm.texture_stats(raw, lab)
    for p in TEXTURE_PROPS:
        np.testing.assert_allclose(tx[p], ref[p], rtol=1e-9, atol=1e-12, equal_nan=True, err_msg=f"seed {seed} {p}")
