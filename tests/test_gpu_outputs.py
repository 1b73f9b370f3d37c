"""GPU tests of the GeoDataFrame / GeoPackage-out surface north_star names: the table create_segments returns
(segment_boundaries.py:59-77), the objects table of segment() with the reference's default columns (segment.py:63-93),
and `segments.gpkg` written by create_tiled_segments(output_dir=...) (tiling.py:289-291) -- read back and compared with
the label raster: ids 1..N, polygon area == pixel count x pixel area, one polygon per id."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


class FakeImage:
    """What the path needs of obia.handlers.geotif.Image (geotif.py:8-44)."""
    def __init__(self, img_data, affine_transformation=None, crs=None):
        self.img_data = img_data
        self.affine_transformation = affine_transformation
        self.crs = crs


def synth(H, W, C, seed=0):
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    return np.stack([400 * np.sin(xx / (11 + 3 * c)) * np.cos(yy / (13 + 2 * c)) + 1000 + 50 * c + rs.normal(0, 20, (H, W))
                     for c in range(C)], -1).astype(np.float32)


def ring_area(r):
    x, y = r[:, 0], r[:, 1]
    return 0.5 * float(np.sum(x[:-1] * y[1:] - x[1:] * y[:-1]))


def wkb_area(wkb):
    from obia_amd.geopackage import wkb_rings
    return sum(abs(ring_area(p[0])) - sum(abs(ring_area(h)) for h in p[1:]) for p in wkb_rings(wkb))


def geom_bytes(g):
    return bytes(g) if isinstance(g, (bytes, bytearray)) else g.wkb


@pytest.mark.parametrize("entry", ["numpy", "tensor"])
def test_tiled_segments_gpkg_matches_label_raster(tmp_path, entry):
    from obia_amd.tiling import create_tiled_segments
    from obia_amd.geopackage import read_geopackage
    img = synth(230, 300, 4)
    yy, xx = np.mgrid[0:230, 0:300]
    mask = (yy - 110) ** 2 + (xx - 150) ** 2 < 130 ** 2
    aff = [0.5, 0.0, 0.0, -0.5, 4000.0, 9000.0]          # 0.5 m pixels, north-up
    src = torch.as_tensor(img).cuda() if entry == "tensor" else img
    lab, n = create_tiled_segments(src, str(tmp_path), input_mask=mask, tile_size=100, buffer=16, crown_radius=2.5,
                                   pixel_size=(0.5, 0.5), compactness=10.0, affine_transformation=aff, crs="EPSG:32610")
    lab = lab.cpu().numpy() if entry == "tensor" else lab
    wkbs, cols, srs = read_geopackage(str(tmp_path / "segments.gpkg"))
    assert srs == 32610 and cols["segment_id"] == list(range(1, n + 1)) and len(wkbs) == n
    counts = np.bincount(lab.ravel(), minlength=n + 1)[1:]
    areas = np.array([wkb_area(w) for w in wkbs])
    np.testing.assert_allclose(areas, counts * 0.25, rtol=1e-12)     # polygon area == pixels x 0.25 m^2
    # the polygon of id v sits where the raster has v: its envelope equals the bounding box of the pixels, in map coordinates
    from obia_amd.geopackage import wkb_rings
    for v in (1, n // 2, n):
        ys, xs = np.nonzero(lab == v)
        ext = wkb_rings(wkbs[v - 1])[0][0]
        assert ext[:, 0].min() == 4000.0 + 0.5 * xs.min() and ext[:, 0].max() == 4000.0 + 0.5 * (xs.max() + 1)
        assert ext[:, 1].max() == 9000.0 - 0.5 * ys.min() and ext[:, 1].min() == 9000.0 - 0.5 * (ys.max() + 1)


def test_create_segments_table_and_segment_defaults(tmp_path):
    """create_segments(as_table=True) = the reference's return value (geometry + segment_id 1..N); segment() yields the
    reference's default columns (GLCM included: calculate_textural defaults to True, segment_statistics.py:394)."""
    from obia_amd.segmentation import create_segments, segment
    from obia_amd.geopackage import read_geopackage
    raw = synth(96, 120, 3, seed=2)
    image = FakeImage(raw, affine_transformation=[2.0, 0.0, 0.0, -2.0, 100.0, 500.0], crs="EPSG:32605")
    tab = create_segments(image, method="slic", n_segments=40, compactness=5.0, as_table=True)
    lab = create_segments(image, method="slic", n_segments=40, compactness=5.0)
    n = int(lab.max())
    assert list(tab["segment_id"]) == list(range(1, n + 1)) and np.array_equal(tab.attrs["labels"], lab)
    areas = np.array([wkb_area(geom_bytes(g)) for g in tab["geometry"]])
    np.testing.assert_allclose(areas, np.bincount(lab.ravel(), minlength=n + 1)[1:] * 4.0, rtol=1e-12)
    seg = segment(image, method="slic", n_segments=40, compactness=5.0)
    C = 3
    assert list(seg.segments.columns) == ["segment_id"] \
        + [f"b{b}_{s}" for b in range(C) for s in ("mean", "variance", "min", "max", "skewness", "kurtosis")] \
        + [f"b{b}_{s}" for b in range(C) for s in ("contrast", "dissimilarity", "homogeneity", "ASM", "energy", "correlation")] \
        + ["pai", "fhd", "ch", "mean_intensity", "variance_intensity", "geometry"]
    assert len(seg.segments) == n and not seg.segments["b1_contrast"].isna().any()
    # the caller's raster is still raw (statistics are taken from raw values, utils/utils.py:47)
    np.testing.assert_allclose(seg.segments["b0_mean"].to_numpy()[0], raw[lab == 1, 0].astype(np.float64).mean(), rtol=1e-6)
    path = str(tmp_path / "objects.gpkg")
    seg.write_segments(path)
    if not hasattr(seg.segments, "to_file"):          # written by obia_amd.geopackage (no geopandas on this box)
        wkbs, cols, srs = read_geopackage(path)
        assert srs == 32605 and cols["segment_id"] == list(range(1, n + 1))
        np.testing.assert_allclose(cols["b2_max"], seg.segments["b2_max"].to_numpy(), rtol=0)


def test_segment_with_start_label_zero_and_mask_keeps_every_segment():
    """start_label=0: label 0 is a segment like any other (the reference skips only -1, segment_boundaries.py:62-64);
    with a mask the masked pixels become -1 and get no row."""
    from obia_amd.segmentation import segment
    raw = synth(80, 90, 4, seed=4)
    seg0 = segment(FakeImage(raw), n_segments=30, compactness=10.0, start_label=0)
    seg1 = segment(FakeImage(raw), n_segments=30, compactness=10.0, start_label=1)
    assert int(seg0._segments.min()) == 0 and int(seg1._segments.min()) == 1
    assert len(seg0.segments) == len(seg1.segments) == int(seg1._segments.max())
    np.testing.assert_allclose(seg0.segments["b3_mean"].to_numpy(), seg1.segments["b3_mean"].to_numpy(), rtol=0)
    mask = np.ones((80, 90), bool)
    mask[:, :30] = False
    segm = segment(FakeImage(raw), n_segments=30, compactness=10.0, mask=mask)
    lab = segm._segments
    assert (lab[~mask] == -1).all() and len(segm.segments) == len(np.unique(lab[mask]))
