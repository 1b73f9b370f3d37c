"""Randomised parity of the tiled driver against oracle/tiler.py: seeded geometries (raster sizes off every grid, tile sizes,
buffers, crown radii, pixel sizes), 1..9 bands except 3 (no Lab), compactness from 0.25 to 10, masks with holes / empty tiles / thin
pieces.  Bar: the label rasters are IDENTICAL pixel for pixel and the segment counts equal; ids are 1..N -- with one stated exception:
the centroid means of the HIP path are exact sums rounded once, the reference's (and the oracle's) are float32 sums accumulated pixel
by pixel in raster order ("centroid sums"): the colours of a centroid differ by ~1e-6 relative, and at low compactness
(colour-dominated distances) a pixel whose two best candidates tie within that flips.  One such case in ~1 500 random ones so far (seed
172: 4 pixels); a case whose compactness is below 5 may therefore differ in <= 1e-4 of its pixels with equal segment counts, exactly the
bar of the single-raster random parity (tests/test_gpu_random_parity.py) -- and must then ALSO be pixel-identical to the oracle tiler
run with the HIP path's integer centroid sums, so that nothing but that summation can hide in the tolerance.
(oracle/tiler.py itself stays "parity unpinned", DESIGN.md 2: this pins the HIP tile loops on the restatement.)"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def make_case(seed):
    rs = np.random.RandomState(7000 + seed)
    H = int(rs.randint(120, 330))
    W = int(rs.randint(120, 360))
    C = int(rs.choice([1, 2, 4, 5, 8, 9]))
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    img = np.stack([350 * np.sin(xx / (9 + 3 * c)) * np.cos(yy / (12 + 2 * c)) + 900 + 60 * c + rs.normal(0, 22, (H, W))
                    for c in range(C)], -1).astype(np.float32)
    tile = int(rs.choice([64, 80, 100, 128, 150]))
    buf = int(rs.choice([8, 12, 16, 24, 30]))
    kw = dict(tile_size=tile, buffer=buf, crown_radius=float(rs.choice([3, 4, 5, 6])),
              pixel_size=(float(rs.choice([0.5, 1.0])),) * 2, compactness=float(rs.choice([0.25, 1.0, 10.0])))
    if seed >= 48 and seed < 172:     # (added in round 4; earlier cases keep their geometry) corner lengths that are not whole pixels:
        # odd buffers, pixel sizes that do not divide buffer / 2, different in x and y (tiling.py:189-231)
        buf = kw["buffer"] = int(rs.choice([9, 13, 15, 25]))
        kw["pixel_size"] = (float(rs.choice([0.3, 0.7, 1.0])), float(rs.choice([0.3, 0.5, 1.0])))
        kw["crown_radius"] = float(rs.choice([3, 4, 5])) * max(kw["pixel_size"])
    mask = None
    kind = rs.randint(0, 4)
    if seed >= 40:     # (added in round 3; the first forty cases keep their geometry)
        kind = 4
    if kind == 4:      # islands inside white tiles, everything around them masked: a white tile whose polygon no existing segment
        # intersects keeps its mask untouched -- corner squares included (tiling.py:212, 261-262)
        mask = np.zeros((H, W), bool)
        for tj in range(-(-H // tile)):
            for ti in range(-(-W // tile)):
                if (ti + tj) % 2 == 1 and rs.rand() < 0.7:
                    y0, x0 = tj * tile - buf, ti * tile - buf          # the grown window
                    y1, x1 = tj * tile + tile + buf, ti * tile + tile + buf
                    m = int(rs.randint(0, buf + 6))                    # island = window shrunk by m: reaches the corners when m is small
                    mask[max(0, y0 + m):max(0, y1 - m), max(0, x0 + m):max(0, x1 - m)] = True
    elif kind == 1:      # disc with a rectangular hole
        mask = ((yy - H / 2) ** 2 + (xx - W / 2) ** 2 < (0.48 * max(H, W)) ** 2) & ~((abs(yy - H / 3) < H / 9) & (abs(xx - W / 2) < W / 7))
    elif kind == 2:    # a masked corner (empties whole tiles) and a diagonal band
        mask = np.ones((H, W), bool)
        mask[:tile, :tile] = False
        mask &= (abs(xx - yy * W / H) > 6)
    elif kind == 3:    # scattered rectangles
        mask = np.zeros((H, W), bool)
        for _ in range(5):
            y0, x0 = rs.randint(0, H - 20), rs.randint(0, W - 20)
            mask[y0:y0 + rs.randint(20, H // 2), x0:x0 + rs.randint(20, W // 2)] = True
    return img, mask, kw


def check_against_oracle(lab, n, ref, n_ref, kw, what):
    nd = int((lab != ref).sum())
    if kw["compactness"] >= 5.0:
        assert n == n_ref and nd == 0, f"{what}: {nd} px differ, n {n} vs {n_ref}"
    else:   # (see the module docstring: last-bit centroid colours can flip a near-tie where the colour term decides)
        assert n == n_ref and nd <= 1e-4 * lab.size, f"{what}: {nd} px differ, n {n} vs {n_ref}"


def test_the_known_near_tie_case_stays_within_the_stated_bar(oracle):
    """seed 172 (217 x 299 x 4, compactness 0.25, islands mask): four isolated pixels of one white tile come out with a neighbouring
    label -- before connectivity already, in the single-raster operator on that tile's window (tools/debug_tile_stage.py 172); every
    other stage of that tile and every other tile is identical.  The case is kept so that the effect stays visible and bounded."""
    from obia_amd.tiling import create_tiled_segments
    from oracle import tiler
    img, mask, kw = make_case(172)
    ref, n_ref = tiler.create_tiled_segments(img, mask, **kw)
    lab, n = create_tiled_segments(torch.as_tensor(img).cuda(), input_mask=mask, **kw)
    lab = lab.cpu().numpy()
    assert n == n_ref and int((lab != ref).sum()) <= 6
    again, n2 = create_tiled_segments(torch.as_tensor(img).cuda(), input_mask=mask, **kw)
    assert n2 == n and np.array_equal(again.cpu().numpy(), lab)          # (deterministic: the HIP sums do not depend on the order of the atomics)


@pytest.mark.parametrize("seed", range(int(os.environ.get("OBIA_RANDOM_TILER_CASES", "60"))))
def test_random_tiled_case_vs_oracle(oracle, seed):
    from obia_amd.tiling import create_tiled_segments
    from oracle import tiler
    img, mask, kw = make_case(seed)
    ref, n_ref = tiler.create_tiled_segments(img, mask, **kw)
    lab, n = create_tiled_segments(torch.as_tensor(img).cuda(), input_mask=mask, **kw)
    lab = lab.cpu().numpy()
    check_against_oracle(lab, n, ref, n_ref, kw, f"seed {seed} {img.shape} {kw}")
    if kw["compactness"] < 5.0:
        # the tolerance above is for the centroid sums alone.  So that a real regression of a few pixels in the tile loops (seams,
        # corner squares, the tile_any branch) cannot hide in it, the same case must ALSO equal, pixel for pixel, the oracle tiler run
        # with the HIP path's integer centroid sums (oracle.set_sum_mode(1): every other operation stays the reference's).
        # The comparison above is the parity evidence, this one is the regression guard.
        oracle.set_sum_mode(1)
        try:
            ref1, n_ref1 = tiler.create_tiled_segments(img, mask, **kw)
        finally:
            oracle.set_sum_mode(0)
        assert n == n_ref1 and np.array_equal(lab, ref1), f"seed {seed}: {(lab != ref1).sum()} px differ from the oracle tiler with integer sums"
    if mask is not None:
        assert (lab[~mask] == 0).all()
    if n:
        # ids are 1..N.  Every id has pixels -- except in the one situation a label raster cannot hold ("white tile without
        # neighbouring segments", tiling.py:261-262): a segment lying wholly inside a corner square of a white tile whose polygon no
        # segment meets stays in the reference's table while the new segments are drawn over it; it keeps its id and has no pixel left
        # (the oracle's raster, which `lab` equals pixel for pixel, shows the same).  Seen twice in 170 random geometries (seeds 65 and
        # 125: pixel sizes of 0.3 make segments of a few pixels, a dozen of which fit into one corner square).
        ids = np.unique(lab[lab > 0])
        assert ids[0] >= 1 and ids[-1] <= n
        assert np.array_equal(ids, np.unique(ref[ref > 0]))
