"""Edge cases of the HIP path against the oracle: every band-count instantiation (CP = 4, 8, 12, 16), tiny and
degenerate rasters, float64 / strided input, explicit n_segments and non-square pixels in the tiler, rasters smaller
than one tile, masks that empty whole tiles."""
import numpy as np
import pytest

from tests.metrics import adjusted_rand_index, label_disagreement

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def synth(H, W, C, seed=0, period=9.0):
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    return np.stack([400 * np.sin(xx / (period + 2 * c)) * np.cos(yy / (period + 3 + c)) + 1000 + 50 * c + rs.normal(0, 20, (H, W))
                     for c in range(C)], -1).astype(np.float32)


@pytest.mark.parametrize("C", [1, 2, 5, 6, 9, 12, 13, 16])
@pytest.mark.parametrize("comp", [10.0, 0.3])
def test_every_band_count_vs_oracle(oracle, C, comp):
    from obia_amd.segmentation import slic
    from obia_amd.statistics import zonal_stats
    raw = synth(150, 170, C, seed=C)
    kw = dict(n_segments=80, compactness=comp, convert2lab=False)
    lab = slic(raw, _normalize_bands=True, **kw)
    ref, pre, _ = oracle.slic(oracle.normalize(raw), return_all=True, **kw)
    got_pre = slic(torch.as_tensor(raw).cuda(), _normalize_bands=True, _stage="pre", **kw).cpu().numpy()
    assert label_disagreement(got_pre, pre) <= 1e-4
    assert adjusted_rand_index(lab, ref) >= 0.99
    st = zonal_stats(raw, lab)
    chk = oracle.zonal_stats_numpy(raw, lab)
    assert np.array_equal(st["count"], chk["count"])
    np.testing.assert_allclose(st["mean"], chk["mean"], rtol=1e-5)
    np.testing.assert_allclose(st["variance"], chk["variance"], rtol=1e-5, atol=1e-6 * float(raw.max() - raw.min()) ** 2)
    np.testing.assert_array_equal(st["min"], chk["min"].astype(np.float32))
    np.testing.assert_array_equal(st["max"], chk["max"].astype(np.float32))


def test_more_than_16_bands_is_rejected_loudly():
    from obia_amd.segmentation import slic
    with pytest.raises(NotImplementedError):
        slic(synth(32, 32, 17), n_segments=4)


@pytest.mark.parametrize("shape", [(1, 1, 4), (1, 37, 4), (41, 1, 3), (2, 3, 4), (5, 200, 8), (65, 65, 4)])
def test_tiny_and_degenerate_shapes(oracle, shape):
    from obia_amd.segmentation import slic
    H, W, C = shape
    raw = synth(H, W, C, seed=1)
    if H * W == 1:
        raw[0, 0] = np.arange(C)          # a single pixel is a constant band: normalisation is undefined
        with pytest.raises(ValueError):
            slic(raw, n_segments=1, _normalize_bands=True)
        return
    kw = dict(n_segments=max(1, H * W // 30), compactness=1.0, convert2lab=False)
    lab = slic(raw, _normalize_bands=True, **kw)
    ref = oracle.slic(oracle.normalize(raw), **kw)
    assert lab.shape == (H, W)
    assert adjusted_rand_index(lab, ref) >= 0.99 or np.array_equal(lab, ref)


def test_float64_and_strided_inputs_match_contiguous_float32():
    from obia_amd.segmentation import slic, create_segments
    raw = synth(120, 140, 6)
    a = slic(raw, n_segments=60, compactness=0.5, _normalize_bands=True)
    b = slic(raw.astype(np.float64), n_segments=60, compactness=0.5, _normalize_bands=True)
    planar = np.ascontiguousarray(raw.transpose(2, 0, 1)).transpose(1, 2, 0)     # band-planar strides, like obia's fancy-index copy
    assert not planar.flags.c_contiguous
    c = slic(planar, n_segments=60, compactness=0.5, _normalize_bands=True)
    assert np.array_equal(a, b) and np.array_equal(a, c)
    sel = create_segments(raw, segmentation_bands=[4, 1, 2, 0], n_segments=60, compactness=0.5, convert2lab=False)
    d = slic(raw[:, :, [4, 1, 2, 0]], n_segments=60, compactness=0.5, convert2lab=False, _normalize_bands=True)
    assert np.array_equal(sel, d)


def test_tiler_variants_vs_oracle(oracle):
    from obia_amd.tiling import create_tiled_segments
    from oracle import tiler
    img = synth(190, 260, 4, seed=5)
    # raster smaller than one tile in one direction; explicit n_segments; non-square pixels (corner squares differ per axis)
    for kw in (dict(tile_size=200, buffer=20, crown_radius=5, pixel_size=(1.0, 1.0)),
               dict(tile_size=80, buffer=12, n_segments=40, pixel_size=(1.0, 1.0)),
               dict(tile_size=90, buffer=10, crown_radius=3, pixel_size=(0.5, 2.0))):
        ref, n_ref = tiler.create_tiled_segments(img, None, compactness=10.0, **kw)
        lab, n = create_tiled_segments(img, compactness=10.0, **kw)
        assert np.array_equal(lab == 0, ref == 0)
        assert adjusted_rand_index(lab, ref) >= 0.99 and abs(n - n_ref) <= max(1, 0.01 * n_ref)
    # a mask that removes whole tiles and leaves slivers
    mask = np.zeros((190, 260), bool)
    mask[5:60, 10:250] = True
    mask[100:104, :] = True
    kw = dict(tile_size=80, buffer=12, crown_radius=3, pixel_size=(1.0, 1.0))
    ref, n_ref = tiler.create_tiled_segments(img, mask, compactness=10.0, **kw)
    lab, n = create_tiled_segments(img, input_mask=mask, compactness=10.0, **kw)
    assert (lab[~mask] == 0).all()
    assert adjusted_rand_index(lab, ref) >= 0.99 and abs(n - n_ref) <= max(1, 0.03 * n_ref)


def test_tile_with_constant_band_is_skipped_not_fatal():
    """a nodata block (all zeros) makes normalize_band 0/0 in that tile: the reference's slic raises ValueError on
    the NaNs and the tiler prints "empty tile" and goes on (tiling.py:149-150)"""
    from obia_amd.tiling import create_tiled_segments
    img = synth(160, 160, 4, seed=8)
    img[:80, :80, 2] = 0.0
    lab, n = create_tiled_segments(img, tile_size=80, buffer=8, crown_radius=3, pixel_size=(1.0, 1.0))
    assert n > 0 and (lab[:70, :70] == 0).all() and (lab[90:, 90:] > 0).mean() > 0.95


@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("max_iter", [10, 25])
def test_fixed_point_tile_cache_skips_work_and_keeps_labels(masked, max_iter):
    """exit_on_fixed_point replays the cached partial sums of every 64x64 tile whose candidate centroids stood still.
    On a raster of flat fields most tiles converge within a few sweeps: the option must evaluate fewer pixel-sweeps
    (so the replay path really ran) and still return exactly the labels of the full sweeps."""
    from obia_amd import _lib
    from obia_amd.segmentation import slic
    rs = np.random.RandomState(5)
    H, W, C = 700, 900, 4
    fields = rs.uniform(0, 1000, (7, 9, C)).astype(np.float32)
    raw = np.repeat(np.repeat(fields, 100, 0), 100, 1)[:H, :W].copy()
    raw[:350] += rs.normal(0, 5, (350, W, C)).astype(np.float32)       # the upper half keeps moving
    mask = None
    if masked:
        mask = np.ones((H, W), np.uint8)
        mask[100:180, 200:420] = 0
        mask[:, 860:] = 0
    ctx = _lib.Context(0)
    ctx.set_profiling(True)
    kw = dict(n_segments=600, compactness=20.0, max_num_iter=max_iter, convert2lab=False, mask=mask, _normalize_bands=True, ctx=ctx)
    full = slic(raw, **kw)
    t_full = ctx.timing()
    fast = slic(raw, exit_on_fixed_point=True, **kw)
    t_fast = ctx.timing()
    assert np.array_equal(full, fast)
    px_full = t_full["assign_px"] + t_full["prepass_px"]
    px_fast = t_fast["assign_px"] + t_fast["prepass_px"]
    assert px_fast < 0.9 * px_full, (px_fast, px_full)


@pytest.mark.parametrize("C,bands", [(5, [4, 0, 2]), (9, [8, 0, 7, 3, 5]), (13, [12, 1, 0, 11, 6, 5, 4, 9, 2]),
                                     (16, list(range(15, -1, -1))), (8, [3]), (4, None)])
@pytest.mark.parametrize("block", [4, 23])
def test_zonal_band_subsets_nans_and_crowded_tiles(oracle, C, bands, block):
    """Band lists that are re-ordered subsets (lanes own band quads of the LIST, not of the raster), NaN pixels per
    band, and -- with 4x4-pixel segments -- 256 labels per 64x64 block, four times the slots of the LDS table, so most
    runs take the overflow path straight to global memory.  Ragged raster (not a multiple of 64)."""
    from obia_amd.statistics import zonal_stats
    rs = np.random.RandomState(C * 10 + block)
    H, W = 150, 203
    raw = (rs.uniform(-50, 4000, (H, W, C))).astype(np.float32)
    raw[rs.rand(H, W, C) < 0.01] = np.nan
    yy, xx = np.mgrid[0:H, 0:W]
    lab = ((yy // block) * ((W + block - 1) // block) + xx // block + 1).astype(np.int32)
    lab[rs.rand(H, W) < 0.02] = 0                      # unlabelled pixels are ignored
    st = zonal_stats(raw, lab, bands=bands)
    sub = raw if bands is None else raw[:, :, bands]
    chk = oracle.zonal_stats_numpy(sub, lab)
    assert np.array_equal(st["count"], chk["count"])
    np.testing.assert_allclose(st["mean"], chk["mean"], rtol=1e-5, equal_nan=True)
    np.testing.assert_allclose(st["variance"], chk["variance"], rtol=1e-5, atol=1e-6 * 4050.0 ** 2, equal_nan=True)
    np.testing.assert_array_equal(st["min"], chk["min"].astype(np.float32))
    np.testing.assert_array_equal(st["max"], chk["max"].astype(np.float32))


@pytest.mark.parametrize("shape", [(1, 1), (1, 37), (41, 1), (2, 3), (64, 64), (65, 129)])
def test_widened_rows_on_degenerate_rasters(oracle, shape):
    """polygon rings, edges, skewness / kurtosis and GLCM texture on rasters of one pixel, one row, one column, exactly
    one tile and one tile plus a bit: same answers as the CPU restatements, no out-of-range access."""
    from obia_amd.polygons import polygonize
    from obia_amd.consumers import slic_edge
    from obia_amd.statistics import zonal_stats, texture_stats, TEXTURE_PROPS
    from oracle.polygons import label_rings
    from oracle.consumers import edge_raster
    from oracle import glcm
    H, W = shape
    rs = np.random.RandomState(H * 131 + W)
    lab = rs.randint(0, 4, (H, W)).astype(np.int32)            # 0 = unlabelled
    raw = rs.uniform(0, 100, (H, W, 3)).astype(np.float32)
    tab = polygonize(lab, start_label=1)
    got = [(int(tab.ring_label[r]), bool(tab.ring_is_hole[r]),
            [(int(x), int(y)) for x, y in tab.xy[tab.ring_offset[r]:tab.ring_offset[r + 1]]]) for r in range(len(tab.ring_label))]
    rings = label_rings(lab, start_label=1)
    order = sorted(range(len(rings)), key=lambda i: (rings[i][0], rings[i][1], i))
    assert got == [rings[i] for i in order]
    np.testing.assert_array_equal(slic_edge(lab), edge_raster(lab).astype(np.float32))
    if lab.max() >= 1:
        st = zonal_stats(raw, lab, moments=True)
        chk = oracle.zonal_stats_numpy(raw, lab)
        assert np.array_equal(st["count"], chk["count"])
        np.testing.assert_allclose(st["skewness"], chk["skewness"], rtol=1e-3, atol=1e-3, equal_nan=True)
        tx = texture_stats(raw, lab)
        ref = glcm.texture_stats(raw, lab)
        for p in TEXTURE_PROPS:
            np.testing.assert_allclose(tx[p], ref[p], rtol=1e-9, atol=1e-12, equal_nan=True, err_msg=p)


def test_widened_rows_without_any_label():
    from obia_amd.polygons import polygonize
    from obia_amd.consumers import slic_edge
    lab = np.zeros((20, 30), np.int32)
    tab = polygonize(lab, start_label=1)
    assert len(tab) == 0 and len(tab.ring_label) == 0 and tab.geojson_features() == []
    assert not slic_edge(lab).any()


def test_coarse_segmentation_of_a_big_raster_keeps_64_bit_coordinate_sums():
    """8192 x 8192 with n_segments=100: a cluster holds ~670 000 pixels and the sum of their row indices passes 2^32
    (670 000 x 7 700).  The accumulator records carry n, sum_y and sum_x as separate 64-bit words; with a 32-bit sum_y the
    centroids of the lower rows would land ~6 000 rows too high and the bottom of the raster would come out mislabelled.
    Property: compactness 100 on a smooth image gives a near-regular 10 x 10 grid -- every segment's bounding box is about
    one grid step wide and tall, its centre of mass sits in the middle of the box, and the sizes are within 25 % of the mean."""
    from obia_amd.segmentation import slic
    H = W = 8192
    yy = torch.arange(H, device="cuda", dtype=torch.float32)[:, None]
    xx = torch.arange(W, device="cuda", dtype=torch.float32)[None, :]
    img = (torch.sin(xx / 700.0) * torch.cos(yy / 900.0)).unsqueeze(-1).contiguous()
    lab = slic(img, n_segments=100, compactness=100.0, _normalize_bands=True)
    n = int(lab.max().item())
    assert n == 100 and int(lab.min().item()) == 1
    S = 819.0
    sizes = torch.bincount(lab.reshape(-1), minlength=n + 1)[1:].to(torch.float64)
    assert float(sizes.min()) >= 0.75 * H * W / n and float(sizes.max()) <= 1.25 * H * W / n
    rows = torch.arange(H, device="cuda", dtype=torch.float64)[:, None].expand(H, W)
    sum_y = torch.zeros(n + 1, dtype=torch.float64, device="cuda").index_add_(0, lab.reshape(-1).to(torch.int64), rows.reshape(-1))[1:]
    cy = (sum_y / sizes).cpu().numpy()
    for v in range(1, n + 1):
        ys = torch.nonzero((lab == v).any(dim=1)).reshape(-1)
        y0, y1 = int(ys.min()), int(ys.max())
        assert y1 - y0 <= 1.4 * S, f"segment {v} spans rows {y0}..{y1}"
        assert abs(cy[v - 1] - 0.5 * (y0 + y1)) <= 0.15 * S


def test_failed_tiled_call_leaves_nothing_running_and_the_context_reusable(monkeypatch):
    """the white tiles' feature pass runs on a side stream beside the black batch and is joined when the white pass
    starts.  A failure in between must not return while that pass is still reading the raster / writing arena memory: the call
    is made to fail right after the black pass (test hook), the raster is dropped at once, and the SAME context must then
    produce exactly what a fresh one does."""
    from obia_amd import _lib
    from obia_amd.tiling import create_tiled_segments
    kw = dict(tile_size=128, buffer=16, crown_radius=3, pixel_size=(1.0, 1.0))
    img = torch.as_tensor(synth(400, 420, 8, seed=21)).cuda()
    ctx = _lib.Context(0)
    monkeypatch.setenv("OBIA_DEBUG_FAIL_AFTER_BLACK", "1")
    with pytest.raises(ValueError, match="forced failure"):
        create_tiled_segments(img.clone(), ctx=ctx, **kw)   # (the clone is freed as soon as the call raises)
    monkeypatch.delenv("OBIA_DEBUG_FAIL_AFTER_BLACK")
    junk = torch.full((400, 420, 8), float("nan"), device="cuda")   # whatever took the freed raster's place
    lab, n = create_tiled_segments(img, ctx=ctx, **kw)
    ref, n_ref = create_tiled_segments(img, ctx=_lib.Context(0), **kw)
    del junk
    assert n == n_ref and torch.equal(lab, ref)


def test_pixel_counters_of_a_call_after_a_fixed_point_call_on_the_same_context():
    """Round 4: the orphan flag and pixel counters of a tiled batch travel to a pinned buffer of their own and are read after the
    connectivity stage's synchronisation.  read_back() once freed that buffer when it grew its own (a 4104-byte read-back: the
    exit_on_fixed_point path), and the next calls read recycled memory.  The counters of a plain call must not depend on what ran
    before it on the context."""
    from obia_amd import _lib
    from obia_amd.tiling import create_tiled_segments
    img = torch.as_tensor(synth(300, 320, 8, seed=31)).cuda()
    ctx = _lib.Context(0)
    ctx.set_profiling(1)
    kw = dict(tile_size=128, buffer=16, crown_radius=3, pixel_size=(1.0, 1.0), ctx=ctx)
    create_tiled_segments(img, **kw)
    a = ctx.timing()
    create_tiled_segments(img, exit_on_fixed_point=True, **kw)
    create_tiled_segments(img, **kw)
    b = ctx.timing()
    assert a["assign_px"] == b["assign_px"] == a["prepass_px"] == b["prepass_px"] > 0
