"""Consumers of the label raster: slic_edge against the CPU restatement (oracle/consumers.py, after
obia/utils/cost.py:21-26,44-48) and the point-in-segment join (obia/utils/utils.py:12-34) on hand-made cases."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("case", ["blocks", "sparse", "dense", "flat", "thin"])
def test_slic_edge_matches_the_restatement(case):
    from obia_amd.consumers import slic_edge
    from oracle.consumers import edge_raster
    rs = np.random.RandomState(2)
    if case == "blocks":
        yy, xx = np.mgrid[0:150, 0:203]
        lab = ((yy // 17) * 13 + xx // 19 + 1).astype(np.int32)
    elif case == "sparse":                    # fewer than 2 % edge pixels: the 98th percentile is 0 -> all zeros
        lab = np.ones((300, 300), np.int32); lab[:10, :10] = 2
    elif case == "dense":                     # more than 98 % edge pixels: the 2nd percentile is 1
        lab = rs.randint(0, 1 << 30, (64, 80)).astype(np.int32)
    elif case == "flat":
        lab = np.full((20, 30), 4, np.int32)
    else:
        lab = rs.randint(0, 3, (1, 500)).astype(np.int32)
    got = slic_edge(lab)
    assert got.dtype == np.float32 and got.shape == lab.shape
    np.testing.assert_array_equal(got, edge_raster(lab).astype(np.float32))
    got_t = slic_edge(torch.as_tensor(lab).cuda())
    np.testing.assert_array_equal(got_t.cpu().numpy(), got)


def test_point_join_assigns_classes_and_reports_mixed_segments():
    from obia_amd.consumers import label_segments, sample_labels
    lab = np.zeros((40, 60), np.int32)
    lab[:20, :30] = 1; lab[:20, 30:] = 2; lab[20:, :30] = 3; lab[20:, 30:] = 4
    aff = [0.5, 0.0, 0.0, -0.5, 1000.0, 2000.0]            # 0.5 m pixels, north up, origin (1000, 2000)
    def xy(row, col):                                     # centre of pixel (row, col) in map coordinates
        return (1000.0 + 0.5 * (col + 0.5), 2000.0 - 0.5 * (row + 0.5))
    pts = [xy(3, 3), xy(10, 20), xy(5, 40), xy(6, 50), xy(30, 5), (0.0, 0.0)]
    cls = ["oak", "oak", "oak", "pine", "gap", "oak"]
    seg = sample_labels(lab, aff, pts, outside=-7)
    assert seg.tolist() == [1, 1, 2, 2, 3, -7]
    labelled, mixed = label_segments(lab, aff, pts, cls)
    assert labelled == {1: "oak", 3: "gap"} and mixed == [2]
    # a rotated / sheared transform goes through the same inverse
    aff2 = [0.4, 0.1, -0.05, -0.5, 10.0, 20.0]
    r, c = 25.3, 41.8
    X = aff2[0] * c + aff2[1] * r + aff2[4]; Y = aff2[2] * c + aff2[3] * r + aff2[5]
    assert sample_labels(lab, aff2, [(X, Y)]).tolist() == [4]


def test_quickstart_example_runs(tmp_path, monkeypatch):
    """examples/quickstart.py: segment -> objects table -> polygons -> tiled driver -> consumers, end to end."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(__file__)), "examples", "quickstart.py")
    spec = importlib.util.spec_from_file_location("quickstart_example", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main()
    out = os.path.join(os.path.dirname(path), "quickstart_segments.geojson")
    assert os.path.getsize(out) > 1000
