"""Label raster -> polygon rings: the HIP pass against the CPU restatement (oracle/polygons.py), against
size-independent properties at full size, and the host-side grouping / export helpers."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rings_as_tuples(tab):
    out = []
    for r in range(len(tab.ring_label)):
        xy = tab.xy[tab.ring_offset[r]:tab.ring_offset[r + 1]]
        out.append((int(tab.ring_label[r]), bool(tab.ring_is_hole[r]), [(int(x), int(y)) for x, y in xy]))
    return out


def oracle_grouped(lab, start_label):
    """oracle rings in the table's order: by label, exterior before holes, then raster order of the smallest corner."""
    from oracle.polygons import label_rings
    rings = label_rings(lab, start_label=start_label)
    idx = sorted(range(len(rings)), key=lambda i: (rings[i][0], rings[i][1], i))
    return [rings[i] for i in idx]


@pytest.mark.parametrize("case", ["rect", "donut", "diag", "inv", "salt", "border"])
def test_rings_equal_the_oracle_on_small_maps(case):
    from obia_amd.polygons import polygonize
    rs = np.random.RandomState(11)
    if case == "rect":
        lab = np.zeros((4, 5), np.int32); lab[1:3, 1:4] = 7
    elif case == "donut":
        lab = np.full((5, 5), 1, np.int32); lab[2, 2] = 2
    elif case == "diag":
        lab = np.array([[3, 0], [0, 3]], np.int32)
    elif case == "inv":
        lab = np.array([[0, 3, 3], [3, 0, 3], [3, 3, 3]], np.int32)
    elif case == "salt":
        lab = rs.randint(-1, 4, (37, 53)).astype(np.int32)
    else:
        lab = np.full((70, 130), 5, np.int32); lab[10:60, 10:120] = 6; lab[20:30, 20:30] = 5; lab[69, 129] = -1
    start = 0 if case == "salt" else 1
    tab = polygonize(lab, start_label=start)
    assert rings_as_tuples(tab) == oracle_grouped(lab, start)


@pytest.mark.parametrize("name", ["c2s_256x256x4_c10", "mask_128x160x4_c10", "ragged_200x333x5"])
def test_rings_of_slic_label_maps(name):
    """Golden SLIC label maps (masked pixels are 0 and get no polygon): identical to the oracle, one exterior ring per
    label, ring areas add up to the pixel counts, and rasterising the rings back returns the label map."""
    from obia_amd.polygons import polygonize
    from oracle.polygons import rasterize_rings
    z = np.load(os.path.join(GOLD, name + ".npz"))
    lab = z["labels"].astype(np.int32)
    tab = polygonize(lab, start_label=1)
    got = rings_as_tuples(tab)
    assert got == oracle_grouped(lab, 1)
    ids, counts = np.unique(lab[lab >= 1], return_counts=True)
    assert np.array_equal(tab.labels, ids)
    area = np.zeros(int(ids.max()) + 1)
    np.add.at(area, tab.ring_label, tab.areas())
    assert np.array_equal(area[ids], counts.astype(np.float64))
    ext = np.bincount(tab.ring_label[~tab.ring_is_hole], minlength=int(ids.max()) + 1)
    assert np.all(ext[ids] == 1)
    assert np.array_equal(rasterize_rings(got, *lab.shape, fill=0), np.where(lab >= 1, lab, 0))


def test_affine_transform_wkb_and_geojson():
    from obia_amd.polygons import polygonize
    lab = np.full((5, 5), 1, np.int32); lab[2, 2] = 2; lab[0, 4] = -1
    aff = [0.5, 0.0, 0.0, -0.5, 1000.0, 2000.0]          # [a, b, d, e, xoff, yoff]: 0.5 m pixels, north up
    tab = polygonize(torch.as_tensor(lab).cuda(), affine_transformation=aff, start_label=0)
    assert list(tab.labels) == [1, 2] and len(tab) == 2
    feats = tab.geojson_features()
    assert [f["properties"]["segment_id"] for f in feats] == [1, 2]
    g1 = feats[0]["geometry"]
    assert g1["type"] == "Polygon" and len(g1["coordinates"]) == 2          # exterior + the hole around label 2
    assert g1["coordinates"][0][0] == [1000.0, 2000.0]
    assert feats[1]["geometry"]["coordinates"][0][:2] == [[1001.0, 1999.0], [1001.5, 1999.0]]
    wkb = tab.wkb()
    import struct
    order, gtype, nrings = struct.unpack("<BII", wkb[0][:9])
    assert (order, gtype, nrings) == (1, 3, 2)
    npts = struct.unpack("<I", wkb[0][9:13])[0]
    assert npts == len(g1["coordinates"][0])
    # |signed area| in map units: 0.25 m^2 per pixel
    np.testing.assert_allclose(np.abs(tab.areas()), [0.25 * 24, 0.25, 0.25])   # exterior of 1 (24 px + the hole), hole, label 2


def test_label_with_two_parts_becomes_a_multipolygon():
    from obia_amd.polygons import polygonize
    lab = np.zeros((8, 12), np.int32)
    lab[1:7, 1:6] = 4; lab[3, 3] = 0            # part A with a hole
    lab[2:5, 8:11] = 4                          # part B, not connected to A
    tab = polygonize(lab, start_label=1)
    f = tab.geojson_features()[0]
    assert f["geometry"]["type"] == "MultiPolygon"
    parts = f["geometry"]["coordinates"]
    assert sorted(len(p) for p in parts) == [1, 2]          # the hole went to the part that contains it


def test_full_size_properties():
    """BASELINE configs[1] (4096^2, n = 50 000): every label has one exterior ring, the signed ring areas add up to the
    pixel counts, rings are closed, and the whole pass runs in one call."""
    from obia_amd.segmentation import slic
    from obia_amd.polygons import polygonize
    H = W = 4096
    g = torch.Generator(device="cuda").manual_seed(0)
    yy = torch.arange(H, device="cuda", dtype=torch.float32)[:, None]
    xx = torch.arange(W, device="cuda", dtype=torch.float32)[None, :]
    img = torch.stack([400.0 * torch.sin(xx / (11 + 3 * c)) * torch.cos(yy / (13 + 2 * c)) + 1000 + 50 * c
                       + 20.0 * torch.randn((H, W), device="cuda", generator=g) for c in range(4)], dim=-1)
    lab = slic(img, n_segments=50000, compactness=10.0, _normalize_bands=True)
    tab = polygonize(lab, start_label=1)
    n = int(lab.max().item())
    counts = torch.bincount(lab.flatten().to(torch.int64), minlength=n + 1).cpu().numpy()
    assert len(tab) == n
    area = np.zeros(n + 1)
    np.add.at(area, tab.ring_label, tab.areas())
    assert np.array_equal(area[1:], counts[1:].astype(np.float64))
    ext = np.bincount(tab.ring_label[~tab.ring_is_hole], minlength=n + 1)
    assert np.all(ext[1:] == 1)
    first = tab.xy[tab.ring_offset[:-1]]
    last = tab.xy[tab.ring_offset[1:] - 1]
    assert np.array_equal(first, last)
