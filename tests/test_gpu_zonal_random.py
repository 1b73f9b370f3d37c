"""Randomised parity of the zonal-statistics operator (B2, a9-a10: crop / mask / calculate_spectral_stats) against the NumPy
restatement (oracle.zonal_stats_numpy): seeded rasters of 1..16 bands (every lane layout of the kernel: band quads, partial last quad, band triples for 3 / 6 / 9, gathered
subsets) with NaN pixels, label maps with gaps in the numbering,
labels outside [start_label, start_label + N) (the -1 / 0 of masked pixels), thin and large segments, band subsets in any order,
raster sizes off the 128 x 64 tile of the kernel.  Bar: counts, min and max exact; mean and variance within 1e-5 relative
(variance + 1e-6 range^2), the same NaN pattern."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def make_case(seed):
    rs = np.random.RandomState(11000 + seed)
    H, W, C = int(rs.randint(3, 210)), int(rs.randint(3, 260)), int(rs.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 13, 16]))
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    raw = np.stack([rs.uniform(50, 500) * np.sin(xx / (7 + c)) * np.cos(yy / (9 + c)) + rs.uniform(-200, 2000) + rs.normal(0, 30, (H, W))
                    for c in range(C)], -1).astype(np.float32)
    if rs.rand() < 0.6:
        raw[rs.rand(H, W, C) < rs.choice([0.001, 0.02, 0.3])] = np.nan
    if rs.rand() < 0.2:
        raw[:, :, rs.randint(0, C)] = np.nan            # a whole band without data
    s = int(rs.choice([2, 5, 11, 40]))
    lab = ((yy + 2 * rs.randn(H, W)) // s).astype(np.int64) * ((W + s - 1) // s + 3) + ((xx + 2 * rs.randn(H, W)) // s).astype(np.int64)
    lab = lab - lab.min() + 1
    if rs.rand() < 0.5:                                    # gaps in the numbering: every third label removed
        lab[lab % 3 == 0] = 0
    if rs.rand() < 0.5:
        lab[rs.rand(H, W) < 0.05] = -1
    start_label = int(rs.choice([0, 1]))
    if start_label == 0:
        lab = np.where(lab > 0, lab - 1, -1)
    bands = None
    if C > 1 and rs.rand() < 0.5:
        bands = [int(b) for b in rs.permutation(C)[:rs.randint(1, C + 1)]]
    return raw, lab.astype(np.int32), start_label, bands


@pytest.mark.parametrize("seed", range(int(os.environ.get("OBIA_RANDOM_ZONAL_CASES", "60"))))
def test_random_zonal_case_vs_numpy(oracle, seed):
    from obia_amd.statistics import zonal_stats
    raw, lab, start_label, bands = make_case(seed)
    ref = oracle.zonal_stats_numpy(raw, lab, bands=bands, start_label=start_label)
    st = zonal_stats(torch.as_tensor(raw).cuda(), torch.as_tensor(lab).cuda(), bands=bands, start_label=start_label)
    st = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in st.items()}
    assert np.array_equal(st["count"], ref["count"]), f"seed {seed}"
    for k in ("mean", "variance", "min", "max"):
        assert np.array_equal(np.isnan(st[k]), np.isnan(ref[k])), f"seed {seed}: NaN pattern of {k}"
    fin = np.isfinite(raw)
    rng = float(raw[fin].max() - raw[fin].min()) if fin.any() else 1.0
    np.testing.assert_allclose(st["mean"], ref["mean"], rtol=1e-5, atol=1e-7 * rng, equal_nan=True)
    np.testing.assert_allclose(st["variance"], ref["variance"], rtol=1e-5, atol=1e-6 * rng * rng, equal_nan=True)
    assert np.array_equal(np.nan_to_num(st["min"], nan=-1.0), np.nan_to_num(ref["min"].astype(np.float32), nan=-1.0))
    assert np.array_equal(np.nan_to_num(st["max"], nan=-1.0), np.nan_to_num(ref["max"].astype(np.float32), nan=-1.0))
