"""Randomised parity of quickshift against the oracle's restatement of scikit-image 0.18.3's _quickshift_cy:
ragged shapes (down to a few pixels, narrower than the window), 1..8 bands, kernel sizes on both sides of the LDS-staged window,
max_dist below and above the window radius, ratio.  The density sums are float64 in a fixed order on both sides; the bar is the one
of the fixed cases (ARI >= 0.99, segment count within 2 %), plus consecutive ids and a bit-identical second run."""
import os

import numpy as np
import pytest

from tests.metrics import adjusted_rand_index

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("seed", range(int(os.environ.get("OBIA_RANDOM_QS_CASES", "16"))))
def test_random_quickshift_vs_oracle(oracle, seed):
    from obia_amd.segmentation import quickshift
    rs = np.random.RandomState(23000 + seed)
    H, W, C = int(rs.randint(3, 120)), int(rs.randint(3, 140)), int(rs.choice([1, 2, 3, 3, 4, 8]))
    ks = float(rs.choice([1.0, 2.0, 3.0, 4.5, 5.0, 7.0]))
    md = float(rs.choice([2.0, 6.0, 10.0, 25.0]))
    ratio = float(rs.choice([0.3, 1.0]))
    yy, xx = np.mgrid[0:H, 0:W]
    by, bx = int(rs.randint(5, 40)), int(rs.randint(5, 40))
    base = np.stack([((yy // by + xx // bx + c) % 3) / 2.0 for c in range(C)], -1)
    img = np.clip(base + 0.04 * rs.normal(size=base.shape), 0, 1).astype(np.float32)
    noise = np.random.RandomState(seed).normal(scale=0.00001, size=(H, W))
    ref = oracle.quickshift_core(img.astype(np.float64) * ratio, noise, ks, md)
    lab = quickshift(img, ratio=ratio, kernel_size=ks, max_dist=md, convert2lab=False, random_seed=seed)
    n, n_ref = len(np.unique(lab)), len(np.unique(ref))
    assert lab.shape == (H, W) and lab.min() == 0 and lab.max() == n - 1
    assert abs(n - n_ref) <= max(1, 0.02 * n_ref), f"seed {seed}: {n} segments, oracle {n_ref} ({H}x{W}x{C}, ks {ks}, md {md})"
    assert adjusted_rand_index(lab, ref) >= 0.99, f"seed {seed} ({H}x{W}x{C}, ks {ks}, md {md}, ratio {ratio})"
    again = quickshift(torch.as_tensor(img).cuda(), ratio=ratio, kernel_size=ks, max_dist=md, convert2lab=False, random_seed=seed)
    assert np.array_equal(again.cpu().numpy(), lab)
