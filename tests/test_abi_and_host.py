"""CPU-side checks (no GPU, no compute calls): the C-ABI library loads and exports every symbol that
include/obia_hip.h declares, the ctypes structs match the header, and the host-side mirror of the reference
interface rejects bad arguments before touching the device."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "obia_hip.h")


@pytest.fixture(scope="module")
def lib():
    from obia_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(obia_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 15
    cdll = ctypes.CDLL(lib.LIB_PATH)
    missing = [s for s in syms if not hasattr(cdll, s)]
    assert not missing, f"declared in include/obia_hip.h but not exported: {missing}"
    assert set(syms) == set(lib.EXPORTED_SYMBOLS), "ctypes binding table out of sync with the header"
    assert lib.load().obia_abi_version() == 2


def test_param_structs_match_header(lib):
    p = lib.SlicParams()
    lib.load().obia_slic_default_params(ctypes.byref(p))
    assert (p.n_segments, p.compactness, p.max_num_iter, p.convert2lab, p.enforce_connectivity) == (100, 10.0, 10, -1, 1)
    assert (p.min_size_factor, p.max_size_factor, p.slic_zero, p.start_label, p.normalize_bands) == (0.5, 3.0, 0, 1, 0)
    assert ctypes.sizeof(lib.SlicParams) == 3 * 8 + 10 * 4 + 6 * 8          # 3 doubles, 9 int32 + one reserved, sigma_zyx[3], spacing_zyx[3] (ABI 2)
    assert p.exit_on_fixed_point == 0 and p.reserved == 0 and list(p.sigma_zyx) == [0.0, 0.0, 0.0] and list(p.spacing_zyx) == [1.0, 1.0, 1.0]
    assert ctypes.sizeof(lib.TilingParams) == 3 * 8 + 4 * 4


def test_no_gpu_means_loud_failure_not_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU path|no HIP device"):
        lib.Context(0)
    from obia_amd.segmentation import slic
    with pytest.raises(RuntimeError):
        slic(np.zeros((8, 8, 4), np.float32), n_segments=4)


def test_host_argument_checks_happen_before_device_use():
    from obia_amd.segmentation import slic, create_segments
    from obia_amd.tiling import create_tiled_segments
    img = np.zeros((8, 8, 4), np.float32)
    with pytest.raises(ValueError):
        slic(img, start_label=3)
    with pytest.raises(ValueError):
        slic(img, sigma=-2)            # (sigma >= 0 is supported since round 3: tests/test_gpu_sigma.py)
    with pytest.raises(ValueError):
        slic(img, sigma=[1, 2, 3, 4])   # (two elements are the (row, column) form of scikit-image >= 0.19)
    with pytest.raises(ValueError):
        slic(img, spacing=(1, -1, 1))
    with pytest.raises(IndexError):
        create_segments(img, segmentation_bands=[7])
    with pytest.raises(Exception, match="unknown segmentation method"):
        create_segments(img, method="felzenszwalb")
    with pytest.raises(TypeError):
        create_segments(img, bogus_kwarg=1)
    with pytest.raises(ValueError, match="only the 'slic' method"):
        create_tiled_segments(img, method="quickshift")


def test_objects_table_columns_follow_reference_order():
    from obia_amd.statistics import stats_columns
    cols = stats_columns([0, 2], [1])
    assert cols[:7] == ["segment_id", "b0_mean", "b0_variance", "b0_min", "b0_max", "b0_skewness", "b0_kurtosis"]
    assert cols[7:13] == ["b2_mean", "b2_variance", "b2_min", "b2_max", "b2_skewness", "b2_kurtosis"]
    assert cols[13:] == ["b1_contrast", "b1_dissimilarity", "b1_homogeneity", "b1_ASM", "b1_energy", "b1_correlation"]
    assert stats_columns([0], [], calc_skewness=False, calc_kurtosis=False) == ["segment_id", "b0_mean", "b0_variance", "b0_min", "b0_max"]


def test_objects_table_schema_equals_the_reference_source():
    """create_objects' positional order, defaults and default column list against the reference's own source text
    (obia/segmentation/segment_statistics.py:392-398 and :66-108), parsed with ast when /root/reference is present (it is
    not on the GPU box; the expected values are also written out below)."""
    import ast
    import inspect
    from obia_amd.statistics import create_objects, stats_columns
    sig = inspect.signature(create_objects)
    pos = [p.name for p in sig.parameters.values() if p.kind == p.POSITIONAL_OR_KEYWORD]
    expected = ["segments", "image", "ept", "ept_srs", "spectral_bands", "textural_bands", "voxel_resolution",
                "calculate_spectral", "calculate_textural", "calculate_structural", "calculate_radiometric",
                "calc_mean", "calc_variance", "calc_min", "calc_max", "calc_skewness", "calc_kurtosis", "calc_contrast",
                "calc_dissimilarity", "calc_homogeneity", "calc_ASM", "calc_energy", "calc_correlation", "calc_pai", "calc_fhd",
                "calc_ch", "calc_mean_intensity", "calc_variance_intensity"]
    assert pos == expected
    defaults = {p.name: p.default for p in sig.parameters.values() if p.kind == p.POSITIONAL_OR_KEYWORD and p.default is not p.empty}
    assert defaults["calculate_textural"] is True and defaults["calculate_structural"] is False and defaults["calc_pai"] is True
    ref = "/root/reference/obia/segmentation/segment_statistics.py"
    import os
    if os.path.exists(ref):
        fn = [n for n in ast.parse(open(ref).read()).body if isinstance(n, ast.FunctionDef) and n.name == "create_objects"][0]
        names = [a.arg for a in fn.args.args]
        dflt = [ast.literal_eval(d) for d in fn.args.defaults]
        assert names == expected
        assert dict(zip(names[len(names) - len(dflt):], dflt)) == defaults
    # default columns for a 2-band raster: what segment() yields in the reference (segment.py:87-91 passes no textural flag)
    cols = stats_columns([0, 1], [0, 1], calc_pai=True, calc_fhd=True, calc_ch=True, calc_mean_intensity=True,
                         calc_variance_intensity=True, geometry=True)
    assert cols == ["segment_id"] + [f"b{b}_{s}" for b in (0, 1) for s in ("mean", "variance", "min", "max", "skewness", "kurtosis")] \
        + [f"b{b}_{s}" for b in (0, 1) for s in ("contrast", "dissimilarity", "homogeneity", "ASM", "energy", "correlation")] \
        + ["pai", "fhd", "ch", "mean_intensity", "variance_intensity", "geometry"]


def test_geopackage_round_trip(tmp_path):
    """obia_amd.geopackage writes what tiling.py:291 writes through GDAL: a GeoPackage with `segments` (geometry +
    segment_id).  Round trip of a polygon with a hole and a multipolygon; container magic, metadata rows, blob headers."""
    import sqlite3
    import struct
    from obia_amd.geopackage import write_geopackage, read_geopackage, wkb_rings

    def poly(rings):
        b = struct.pack("<BII", 1, 3, len(rings))
        for r in rings:
            b += struct.pack("<I", len(r)) + np.asarray(r, "<f8").tobytes()
        return b
    sq = lambda x0, y0, s: [(x0, y0), (x0 + s, y0), (x0 + s, y0 + s), (x0, y0 + s), (x0, y0)]      # noqa: E731
    a = poly([sq(0, 0, 10), sq(2, 2, 3)])
    b = struct.pack("<BII", 1, 6, 2) + poly([sq(20, 0, 4)]) + poly([sq(30, 5, 2)])
    path = write_geopackage(str(tmp_path / "segments.gpkg"), [a, b], {"segment_id": [1, 2], "b0_mean": [0.5, float("nan")]},
                            srs_epsg=32610)
    wkbs, cols, srs = read_geopackage(path)
    assert wkbs == [a, b] and cols["segment_id"] == [1, 2] and cols["b0_mean"] == [0.5, None] and srs == 32610
    assert [len(p) for p in wkb_rings(wkbs[1])] == [1, 1] and len(wkb_rings(wkbs[0])[0]) == 2
    con = sqlite3.connect(path)
    assert con.execute("PRAGMA application_id").fetchone()[0] == 0x47504B47
    assert con.execute("SELECT min_x, min_y, max_x, max_y FROM gpkg_contents").fetchone() == (0.0, 0.0, 32.0, 10.0)
    assert con.execute("SELECT count(*) FROM gpkg_spatial_ref_sys WHERE srs_id IN (-1, 0, 4326, 32610)").fetchone()[0] == 4
    blob = con.execute("SELECT geom FROM segments WHERE fid = 2").fetchone()[0]
    assert blob[:2] == b"GP" and struct.unpack_from("<i", blob, 4)[0] == 32610
    assert struct.unpack_from("<4d", blob, 8) == (20.0, 32.0, 0.0, 7.0)
    # the registered geometry type follows the blobs (a POLYGON table that holds a MultiPolygon is rejected by
    # strict readers): a mix is GEOMETRY, polygons only POLYGON, multipolygons only MULTIPOLYGON; a WKT definition is carried
    assert con.execute("SELECT geometry_type_name FROM gpkg_geometry_columns").fetchone()[0] == "GEOMETRY"
    con.close()
    for wk, want in (([a, a], "POLYGON"), ([b], "MULTIPOLYGON")):
        p2 = write_geopackage(str(tmp_path / "t.gpkg"), wk, {"segment_id": list(range(1, len(wk) + 1))}, srs_epsg=32610,
                              srs_wkt='PROJCS["WGS 84 / UTM zone 10N"]')
        con = sqlite3.connect(p2)
        assert con.execute("SELECT geometry_type_name FROM gpkg_geometry_columns").fetchone()[0] == want
        assert con.execute("SELECT definition FROM gpkg_spatial_ref_sys WHERE srs_id = 32610").fetchone()[0].startswith("PROJCS")
        con.close()


def test_oracle_tiler_properties(oracle):
    """The CPU restatement of the tile loops (test infrastructure): ids are 1..N, every segment is 4-connected,
    masked pixels stay 0, the crown rule sets the density."""
    from oracle import tiler
    from scipy import ndimage
    rs = np.random.RandomState(1)
    H, W = 210, 250
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    img = np.stack([400 * np.sin(xx / (11 + 3 * c)) * np.cos(yy / (13 + 2 * c)) + 1000 + rs.normal(0, 20, (H, W))
                    for c in range(4)], -1).astype(np.float32)
    mask = np.ones((H, W), bool)
    mask[:40, :60] = False
    lab, n = tiler.create_tiled_segments(img, mask, tile_size=100, buffer=16, crown_radius=5, pixel_size=(1.0, 1.0))
    ids = np.unique(lab[lab > 0])
    assert ids[0] == 1 and ids[-1] == n == len(ids)
    assert (lab[~mask] == 0).all()
    expected = mask.sum() / (np.pi * 25)
    assert 0.6 * expected <= n <= 1.4 * expected
    st = ndimage.generate_binary_structure(2, 1)
    for v in ids[::17]:
        assert ndimage.label(lab == v, st)[1] == 1
