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


def test_polygon_rings_known_answers():
    """Hand-made label maps: a rectangle, an L shape, a ring with a hole (and the hole's own polygon), and two pixels of
    one label that touch only diagonally (4-connectivity: two separate rings, as GDAL polygonize gives by default,
    segment_boundaries.py:66)."""
    from oracle.polygons import label_rings, rasterize_rings
    lab = np.zeros((4, 5), np.int32)
    lab[1:3, 1:4] = 7
    r = label_rings(lab, start_label=1)
    assert r == [(7, False, [(1, 1), (4, 1), (4, 3), (1, 3), (1, 1)])]
    lab = np.array([[1, 1, 2],
                    [1, 2, 2]], np.int32)
    r = label_rings(lab, start_label=1)
    assert r[0] == (1, False, [(0, 0), (2, 0), (2, 1), (1, 1), (1, 2), (0, 2), (0, 0)])
    assert r[1] == (2, False, [(2, 0), (3, 0), (3, 2), (1, 2), (1, 1), (2, 1), (2, 0)])
    donut = np.full((5, 5), 1, np.int32)
    donut[2, 2] = 2
    r = label_rings(donut, start_label=1)
    assert r[0] == (1, False, [(0, 0), (5, 0), (5, 5), (0, 5), (0, 0)])
    assert r[1] == (2, False, [(2, 2), (3, 2), (3, 3), (2, 3), (2, 2)])        # raster order of the smallest corner:
    assert r[2] == (1, True, [(2, 2), (2, 3), (3, 3), (3, 2), (2, 2)])         # exterior of 2 before the hole of 1
    diag = np.array([[3, 0],
                     [0, 3]], np.int32)
    r = label_rings(diag, start_label=1)
    assert [x[2] for x in r] == [[(0, 0), (1, 0), (1, 1), (0, 1), (0, 0)], [(1, 1), (2, 1), (2, 2), (1, 2), (1, 1)]]
    # background pixels that touch diagonally: the label's outline hugs the label at that corner (its two pixels there
    # are not joined through the corner), so the enclosed background pixel is reached by the exterior ring -- one ring
    # that passes the corner (1, 1) twice, no hole
    inv = np.array([[0, 3, 3],
                    [3, 0, 3],
                    [3, 3, 3]], np.int32)
    r = label_rings(inv, start_label=1)
    assert len(r) == 1 and r[0][1] is False and r[0][2].count((1, 1)) == 2
    for m in (lab, donut, diag, inv):
        back = rasterize_rings(label_rings(m, start_label=1), *m.shape, fill=0)
        assert np.array_equal(back, np.where(m >= 1, m, 0))


def test_polygon_rings_properties_on_random_label_maps():
    from oracle.polygons import label_rings, rasterize_rings
    rs = np.random.RandomState(3)
    for trial in range(5):
        H, W = rs.randint(5, 30), rs.randint(5, 30)
        lab = rs.randint(-1, 4, (H, W)).astype(np.int32)          # salt-and-pepper: many holes and diagonal contacts
        rings = label_rings(lab, start_label=0)
        area = {}
        for L, hole, verts in rings:
            a2 = sum(x0 * y1 - x1 * y0 for (x0, y0), (x1, y1) in zip(verts[:-1], verts[1:]))
            assert (a2 < 0) == hole and verts[0] == verts[-1]
            area[L] = area.get(L, 0) + a2
        for L in np.unique(lab[lab >= 0]):
            assert area[int(L)] == 2 * int((lab == L).sum())
        assert np.array_equal(rasterize_rings(rings, H, W, fill=-1), np.where(lab >= 0, lab, -1))


def test_edge_raster_known_answers():
    """slic_edge (obia/utils/cost.py:44-48) on hand-made maps: an edge pixel is one whose lower or right neighbour has
    another label; the percentile stretch of a 0/1 raster leaves it unchanged when between 2 % and 98 % of the pixels
    are edges, and zeroes it when fewer than 2 % are (98th percentile 0 -> 0/0 -> 0)."""
    from oracle.consumers import edge_raster, percentile_stretch
    lab = np.array([[1, 1, 2],
                    [1, 3, 3],
                    [1, 3, 3]])
    assert edge_raster(lab).tolist() == [[0, 1, 1], [1, 0, 0], [1, 0, 0]]
    big = np.ones((100, 100), int); big[0, 0] = 2
    assert not edge_raster(big).any()
    v = np.arange(101, dtype=np.float64)
    out = percentile_stretch(v)
    assert out[0] == 0 and out[2] == 0 and out[100] == 1 and out[98] == 1 and abs(out[50] - 0.5) < 1e-12


def test_tiler_corner_squares_that_cut_through_pixels():
    """tiling.py:189-231 with a corner length that is not a whole number of pixels (odd buffer, or a pixel size that does not divide
    buffer / 2): rasterize() burns the pixels whose CENTRE is inside the square, a segment whose pixels lie WHOLLY inside has no area
    in tile_polygon (not selected), and a segment with a pixel the square reaches at all is not `within` (Weak 3).
    Known answers for the three counts, and the segment the cut passes through: `overlaps` -> kept and masked, not dropped."""
    from oracle.tiler import OracleTiler
    rs = np.random.RandomState(3)
    img = rs.rand(40, 40, 2).astype(np.float32)
    t = OracleTiler(img, None, 40, 0, tile_size=20, buffer=5, crown_radius=2, pixel_size=(1.0, 1.0))
    assert (t.clx, t.clx_in, t.clx_any) == (2, 2, 3) and (t.cly, t.cly_in, t.cly_any) == (2, 2, 3)        # a = 2.5
    u = OracleTiler(img, None, 40, 0, tile_size=20, buffer=5, crown_radius=2, pixel_size=(0.7, 0.5))
    assert (u.clx, u.clx_in, u.clx_any) == (4, 3, 4) and (u.cly, u.cly_in, u.cly_any) == (5, 5, 5)        # a = 3.571..., 5
    v = OracleTiler(img, None, 40, 0, tile_size=20, buffer=12, crown_radius=2, pixel_size=(0.5, 1.0))
    assert (v.clx, v.clx_in, v.clx_any) == (12, 12, 12) and (v.cly, v.cly_in, v.cly_any) == (6, 6, 6)     # whole numbers: one count
    # white tile (0, 1): window rows [0, 25), columns [15, 40).  Segment 1 = the one pixel at window (24, 2): the square (side 2.5)
    # covers half of it -- it has area in the polygon and is not within it: overlaps.  Segment 2 = window (24, 0): wholly inside
    # the square -- not selected, but under the burned corner.  Segment 3 = window (10, 10): within -> dropped, re-segmented.
    for g, (y, x) in enumerate([(24, 15 + 2), (24, 15 + 0), (10, 15 + 10)], start=1):
        t.G[y, x] = g
    t.set_segments(1, [1, 1, 1])
    t.run(True, 0, 1)
    assert t.alive[1] and t.G[24, 17] == 1          # kept (and masked: no new segment took the pixel)
    assert t.alive[2] and t.G[24, 15] == 2          # untouched
    assert not t.alive[3] and t.G[10, 25] > 3       # dropped and re-segmented
