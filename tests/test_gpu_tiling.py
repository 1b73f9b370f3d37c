"""GPU parity of the tiled driver (B3, obia_tiled_slic_f32) against the oracle's restatement of
create_tiled_segments on label rasters (oracle/tiler.py).  The reference's tiler needs GDAL/shapely and has no
fixtures: this stage is "parity unpinned" (DESIGN.md); what is checked is that the HIP tile loops, seam
rules, crown rule, size cuts and id order equal the CPU restatement built on the pinned SLIC oracle -- EXACTLY:
both sides use the same seeding rule and the per-tile SLIC is bit-exact on non-Lab inputs, so the label rasters
must be identical pixel for pixel (the one 3-band case goes through rgb2lab, device powf/cbrtf vs libm, and keeps the
stated ARI / boundary tolerance)."""
import numpy as np
import pytest

from tests.metrics import adjusted_rand_index, boundary_recall_precision

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def synth(H, W, C, seed=0):
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    return np.stack([400 * np.sin(xx / (11 + 3 * c)) * np.cos(yy / (13 + 2 * c)) + 1000 + 50 * c + rs.normal(0, 20, (H, W))
                     for c in range(C)], -1).astype(np.float32)


CASES = [
    dict(H=300, W=340, C=4, tile_size=100, buffer=16, crown_radius=5, pixel_size=(1.0, 1.0), compactness=10.0),
    dict(H=256, W=256, C=8, tile_size=128, buffer=32, crown_radius=5, pixel_size=(0.5, 0.5), compactness=10.0),
    dict(H=230, W=410, C=4, tile_size=100, buffer=20, crown_radius=4, pixel_size=(1.0, 1.0), compactness=0.25),
    dict(H=200, W=200, C=3, tile_size=200, buffer=30, crown_radius=5, pixel_size=(1.0, 1.0), compactness=10.0),  # single tile
    # 8 bands at the author's compactness (notebooks/deepfor.ipynb:402): the colour term decides the boundaries
    dict(H=256, W=300, C=8, tile_size=128, buffer=24, crown_radius=4, pixel_size=(1.0, 1.0), compactness=0.25),
    # ragged segments + a small max_size_factor: the connectivity stage cuts components at max_size inside the tiler
    # (slic's max_size_factor reaches every tile through **kwargs, tiling.py:137-143)
    dict(H=300, W=340, C=4, tile_size=100, buffer=16, crown_radius=5, pixel_size=(1.0, 1.0), compactness=0.05, max_size_factor=1.2),
    dict(H=260, W=260, C=5, tile_size=128, buffer=20, crown_radius=6, pixel_size=(1.0, 1.0), compactness=0.05, max_size_factor=1.05,
         min_size_factor=0.25),
    # tile_size <= 2 * buffer: grown windows of neighbouring white tiles of one row overlap; the driver must fall back to the
    # reference's one-tile-at-a-time order
    dict(H=150, W=170, C=4, tile_size=40, buffer=24, crown_radius=3, pixel_size=(1.0, 1.0), compactness=1.0),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c['H']}x{c['W']}x{c['C']}_t{c['tile_size']}_c{c['compactness']}")
def test_tiled_vs_oracle(oracle, case):
    from obia_amd.tiling import create_tiled_segments
    from oracle import tiler
    c = dict(case)
    img = synth(c.pop("H"), c.pop("W"), c.pop("C"))
    mask = None
    ref, n_ref = tiler.create_tiled_segments(img, mask, **c)
    lab, n = create_tiled_segments(img, input_mask=mask, **c)
    assert lab.shape == ref.shape and lab.dtype == np.int32
    if img.shape[2] == 3:      # Lab: stated tolerance
        assert ((lab == 0) == (ref == 0)).mean() >= 0.9999
        ari = adjusted_rand_index(lab, ref)
        rec, prec = boundary_recall_precision(ref, lab)
        assert ari >= 0.99 and rec >= 0.99 and prec >= 0.99, f"ARI {ari} recall {rec} precision {prec}"
        assert abs(n - n_ref) <= max(1, 0.01 * n_ref)
    else:
        assert n == n_ref and np.array_equal(lab, ref), f"{(lab != ref).sum()} px differ, n {n} vs {n_ref}"
    if "max_size_factor" in c:   # the cut really fires in these cases: without it the partition is another one
        _, n_nocut = tiler.create_tiled_segments(img, mask, **dict(c, max_size_factor=1e9))
        assert n_nocut != n_ref
    ids = np.unique(lab[lab > 0])
    assert ids[0] == 1 and ids[-1] == n and len(ids) == n          # segment_id = 1..N (tiling.py:289-290)


def test_tiled_with_mask_and_empty_tiles(oracle):
    from obia_amd.tiling import create_tiled_segments
    from oracle import tiler
    img = synth(260, 300, 4, seed=3)
    yy, xx = np.mgrid[0:260, 0:300]
    mask = ((yy - 120) ** 2 + (xx - 160) ** 2 < 110 ** 2)
    mask[:100, :100] = False                      # a fully masked black tile: skipped like the reference's "empty tile"
    kw = dict(tile_size=100, buffer=16, crown_radius=5, pixel_size=(1.0, 1.0), compactness=10.0)
    ref, n_ref = tiler.create_tiled_segments(img, mask, **kw)
    lab, n = create_tiled_segments(img, input_mask=mask, **kw)
    assert (lab[~mask] == 0).all()
    assert n == n_ref and np.array_equal(lab, ref), f"{(lab != ref).sum()} px differ, n {n} vs {n_ref}"
    # device-tensor entry gives the identical raster (determinism: integer accumulators, no atomics on floats)
    lab2, n2 = create_tiled_segments(torch.as_tensor(img).cuda(), input_mask=mask, **kw)
    assert n2 == n and np.array_equal(lab2.cpu().numpy(), lab)


def test_white_tile_without_neighbouring_segments_keeps_its_mask(oracle):
    """tiling.py:212 / 261-262: the overlapping segments AND the two corner squares are masked out only when at least one existing
    segment is within / overlaps the tile polygon; a white tile that meets none keeps the mask it read, so its corner squares are
    segmented.  Islands of valid pixels that cover whole grown windows of white tiles (corners included), everything else masked:
    the black tiles around them are empty."""
    from obia_amd.tiling import create_tiled_segments
    from oracle import tiler
    H, W, T, B = 330, 350, 100, 16
    img = synth(H, W, 4, seed=9)
    mask = np.zeros((H, W), bool)
    mask[0:100, 100:200] = True           # white tile (0, 1) itself ...
    mask[100:116, 84:100] = True          # ... and the parts of its grown window that hold its two corner squares (they lie in the
    mask[100:116, 200:216] = True         #     white tiles (1, 0) and (1, 2)); the black tiles around it have no valid pixel
    mask[184:316, 184:316] = True         # a black tile with neighbours, for contrast
    kw = dict(tile_size=T, buffer=B, crown_radius=5, pixel_size=(1.0, 1.0), compactness=10.0)
    t = tiler.OracleTiler(img, mask, H, 0, **kw)
    nty = -(-H // T)
    t.run(False, 0, nty)
    t.run(True, 0, 1)                     # the first white tile row only: tile (0, 1)
    assert (0, 1) in t.untouched_white_tiles, "the case must reach the else branch (tiling.py:261-262)"
    assert (t.G[108:116, 84:92] > 0).all() and (t.G[108:116, 208:216] > 0).all(), "its corner squares are segmented by tile (0, 1)"
    t.run(True, 1, nty)
    ref, n_ref = t.finalize()
    lab, n = create_tiled_segments(img, input_mask=mask, **kw)
    assert n == n_ref and np.array_equal(lab, ref), f"{(lab != ref).sum()} px differ, n {n} vs {n_ref}"
    assert (lab[~mask] == 0).all()


def test_tiled_errors():
    from obia_amd.tiling import create_tiled_segments
    img = synth(64, 64, 4)
    with pytest.raises(ValueError):
        create_tiled_segments(img, method="quickshift")
    with pytest.raises(TypeError):
        create_tiled_segments(img, bogus=1)


def test_corner_squares_that_cut_through_pixels_hip_vs_known_answers():
    """The same three segments as tests/test_oracle_known_answers.py::test_tiler_corner_squares_that_cut_through_pixels, through the
    tiler session of the HIP library (obia_tiler_*): buffer 5 at pixel size 1 -> corner length 2.5 pixels."""
    from obia_amd.distributed import HipTilerEngine
    rs = np.random.RandomState(3)
    img = torch.as_tensor(rs.rand(40, 40, 2).astype(np.float32)).cuda()
    mask = torch.ones((40, 40), dtype=torch.uint8, device="cuda")
    e = HipTilerEngine(img, mask, 40, 0, 20, 5, 2.0, (1.0, 1.0), {}, 8)
    try:
        for g, (y, x) in enumerate([(24, 17), (24, 15), (10, 25)], start=1):
            e.G[y, x] = g
        e.set_segments(1, torch.tensor([1, 1, 1]))
        e.run(True, 0, 1)
        alive = e.get_alive(4).cpu().numpy()
        G = e.G.cpu().numpy()
        assert alive[1] == 1 and G[24, 17] == 1          # the cut passes through it: overlaps -> kept and masked
        assert alive[2] == 1 and G[24, 15] == 2          # wholly inside the square: not selected
        assert alive[3] == 0 and G[10, 25] > 3           # within: dropped, re-segmented
    finally:
        e.close()


@pytest.mark.parametrize("buffer,pixel", [(9, (1.0, 1.0)), (15, (1.0, 1.0)), (12, (0.3, 0.3)), (13, (0.7, 0.5)), (25, (2.0, 3.0))])
def test_tiler_with_a_corner_length_that_is_not_a_whole_number_of_pixels(oracle, buffer, pixel):
    """odd buffers and pixel sizes that do not divide buffer / 2 (Weak 3): HIP tile loops vs oracle/tiler.py"""
    from obia_amd.tiling import create_tiled_segments
    from oracle import tiler
    rs = np.random.RandomState(buffer)
    H, W = 230, 260
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    img = np.stack([350 * np.sin(xx / (9 + 3 * c)) * np.cos(yy / (12 + 2 * c)) + 900 + 60 * c + rs.normal(0, 22, (H, W)) for c in range(4)], -1).astype(np.float32)
    mask = ((yy - H / 2) ** 2 + (xx - W / 2) ** 2 < (0.47 * W) ** 2)
    kw = dict(tile_size=70, buffer=buffer, crown_radius=3.0 * max(pixel), pixel_size=pixel, compactness=10.0)
    ref, n_ref = tiler.create_tiled_segments(img, mask, **kw)
    lab, n = create_tiled_segments(torch.as_tensor(img).cuda(), input_mask=mask, **kw)
    assert n == n_ref and np.array_equal(lab.cpu().numpy(), ref)


def test_tiler_batch_with_orphan_pixels_equals_the_oracle(oracle):
    """A valid pixel that no window reaches keeps the label of the sweep before, which the tiler's sweeps do not store: the batch's
    orphan flag -- read after the connectivity stage since round 4 (slic_run_sweeps mode 1, slic_sweeps_settle) -- makes the batch
    run again with every sweep storing, and the connectivity stage with it.  Islands of a few valid pixels far from every seed
    (farther than two grid steps) produce such pixels in black and in white tiles."""
    from obia_amd.tiling import create_tiled_segments
    from oracle import tiler
    rs = np.random.RandomState(12)
    H, W = 256, 300
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    img = np.stack([350 * np.sin(xx / (9 + 3 * c)) * np.cos(yy / (12 + 2 * c)) + 900 + 60 * c + rs.normal(0, 22, (H, W)) for c in range(4)], -1).astype(np.float32)
    mask = np.zeros((H, W), bool)
    mask[:, :70] = True                       # a block that gets the seeds ...
    mask[10:250:40, 150:152] = True           # ... and islands 80 pixels away from it, two pixels wide
    mask[30:250:40, 260:263] = True
    kw = dict(tile_size=128, buffer=16, crown_radius=6.0, pixel_size=(1.0, 1.0), compactness=10.0)
    from obia_amd import _lib
    ctx = _lib.Context(0)
    ref, n_ref = tiler.create_tiled_segments(img, mask, **kw)
    lab, n = create_tiled_segments(torch.as_tensor(img).cuda(), input_mask=mask, ctx=ctx, **kw)
    lab = lab.cpu().numpy()
    assert ctx.timing()["batch_repeats"] >= 1, "the case is meant to take the repeat path"
    assert n == n_ref and np.array_equal(lab, ref)
    assert (lab[~mask] == 0).all()
