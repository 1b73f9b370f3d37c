"""Randomised parity of the connectivity stage (a6: _enforce_label_connectivity_cython) against the pinned oracle: seeded label maps
with masked (label 0) regions, blobs, stripes and salt-and-pepper noise, sizes off the 64 x 32 tile grid of cc_tile_kernel, min_size
from 1 to a few hundred pixels, max_size from "never reached" down to 1.  Bar: bit-exact labels and counts."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def make_case(seed):
    rs = np.random.RandomState(9000 + seed)
    H, W = int(rs.randint(5, 200)), int(rs.randint(5, 260))
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    kind = rs.randint(0, 4)
    if kind == 0:      # smooth blobs quantised to a few labels
        f = np.sin(xx / rs.uniform(4, 15)) * np.cos(yy / rs.uniform(4, 15)) + 0.3 * rs.randn(H, W)
        lab = np.digitize(f, np.linspace(-1, 1, int(rs.randint(3, 9)))) + 1
    elif kind == 1:    # salt-and-pepper: thousands of tiny components that only touch each other
        lab = rs.randint(1, int(rs.randint(3, 8)), (H, W))
    elif kind == 2:    # grid cells with noisy borders (what a SLIC sweep leaves)
        s = int(rs.randint(5, 20))
        lab = ((yy + 3 * rs.randn(H, W)) // s).astype(np.int64) * 1000 + ((xx + 3 * rs.randn(H, W)) // s).astype(np.int64) + 2000
    else:              # diagonal stripes of width 1..3
        lab = ((xx + yy) // int(rs.randint(1, 4))).astype(np.int64) % int(rs.randint(2, 6)) + 1
    lab = lab.astype(np.int64)
    if rs.rand() < 0.6:   # masked pixels (label start_label - 1 = 0): a disc hole and a band
        hole = (yy - H * rs.uniform(0.2, 0.8)) ** 2 + (xx - W * rs.uniform(0.2, 0.8)) ** 2 < (0.2 * min(H, W)) ** 2
        band = np.abs(xx - yy * W / max(1, H)) < rs.randint(1, 5)
        lab[hole | band] = 0
    # min_size far above the size of EVERY component -- a whole map of "small" components -- used to be left out (thousands of
    # settle rounds from the optimistic side); since round 3 the rounds restart from the pessimistic side: every regime is in.
    mn = int(rs.choice([1, 2, 4, 9, 25, 300] if kind in (0, 2) else [1, 2, 4, 9, 60, 300, 5000]))
    mx = int(rs.choice([H * W + 1, 6 * mn, 3 * mn, mn + 3, max(1, mn // 2), 1]))
    return lab, mn, mx


@pytest.mark.parametrize("noise,min_size", [(4, 300), (2, 300), (2, 30), (3, 12), (6, 1 << 30)])
def test_salt_and_pepper_2048_every_component_small(oracle, noise, min_size):
    """The regime round 2 could not afford: 2048 x 2048 of salt-and-pepper labels, min_size far above (almost) every component --
    hundreds of thousands of small components that only see each other, a handful of surviving ones or none at all.  The reference's
    single raster pass settles this in order; the device iterates the settle times from the pessimistic side (a few dozen rounds)
    after a few optimistic rounds have not converged.  Bit-exact against the oracle, masked band included."""
    from obia_amd.segmentation import enforce_connectivity
    rs = np.random.RandomState(31 + noise)
    H = W = 2048
    lab = rs.randint(1, noise + 1, (H, W)).astype(np.int64)
    lab[700:720, :] = 0                               # a masked band
    lab[1500:1700, 300:900] = noise + 1               # one large component that survives
    ref = oracle.enforce_connectivity(lab, min_size, H * W + 1, start_label=1)
    out, n = enforce_connectivity(torch.as_tensor(lab.astype(np.int32)).cuda(), min_size, H * W + 1, start_label=1)
    out = out.cpu().numpy()
    assert np.array_equal(out, ref), f"{(out != ref).sum()} px differ"
    assert n == len(np.unique(ref[ref > 0]))


@pytest.mark.parametrize("seed", range(int(os.environ.get("OBIA_RANDOM_CC_CASES", "60"))))
def test_random_connectivity_case_vs_oracle(oracle, seed):
    from obia_amd.segmentation import enforce_connectivity
    lab, mn, mx = make_case(seed)
    ref = oracle.enforce_connectivity(lab, mn, mx, start_label=1)
    out, n = enforce_connectivity(torch.as_tensor(lab.astype(np.int32)).cuda(), mn, mx, start_label=1)
    out = out.cpu().numpy()
    assert np.array_equal(out, ref), f"seed {seed}: {(out != ref).sum()} px differ (shape {lab.shape}, min_size {mn}, max_size {mx})"
    assert n == len(np.unique(ref[ref > 0]))


@pytest.mark.parametrize("noise,min_size", [(4, 300), (3, 12)])
def test_settle_times_from_pixel_lists_equal_the_work_list_walks(noise, min_size):
    """Where small components are many the settle times come from per-component pixel lists (cc_settle_eval_kernel) and one walk
    per component (cc_small_target_kernel); the work list of BFS walks (cc_small_bfs_kernel) stays for maps with few small
    components and behind OBIA_CC_WORKLIST.  Same labels either way (the oracle comparison above runs on the default)."""
    from obia_amd.segmentation import enforce_connectivity
    rs = np.random.RandomState(77 + noise)
    H, W = 1024, 1536
    lab = rs.randint(1, noise + 1, (H, W)).astype(np.int32)
    lab[300:310, :] = 0
    lab[600:800, 200:900] = noise + 1
    t = torch.as_tensor(lab).cuda()
    old = os.environ.pop("OBIA_CC_WORKLIST", None)
    try:
        a, na = enforce_connectivity(t, min_size, H * W + 1, start_label=1)
        os.environ["OBIA_CC_WORKLIST"] = "1"
        b, nb = enforce_connectivity(t, min_size, H * W + 1, start_label=1)
    finally:
        os.environ.pop("OBIA_CC_WORKLIST", None)
        if old is not None:
            os.environ["OBIA_CC_WORKLIST"] = old
    assert na == nb and torch.equal(a, b)
