#!/usr/bin/env python3
"""End-to-end walk through the replaced path on one small raster, with the reference's function names:

    segment()                -> label raster + per-segment objects table      (obia.segmentation.segment, segment.py:63-93)
    Segments.polygons()      -> polygons in map coordinates                   (create_segments back half, :59-77)
    create_tiled_segments()  -> the same for a raster that is processed in tiles (obia.utils.tiling, :62-291)
    slic_edge(), label_segments() -> the consumers of the label raster       (utils/cost.py:44-48, utils/utils.py:12-34)

Needs an MI355X (there is no CPU path).  Writes quickstart_objects.csv and quickstart_segments.geojson next to itself.
    python examples/quickstart.py
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from obia_amd import segment, create_objects                       # noqa: E402
from obia_amd.tiling import create_tiled_segments                  # noqa: E402
from obia_amd.consumers import slic_edge, label_segments           # noqa: E402
from obia_amd.polygons import polygonize                           # noqa: E402


class Image:
    """The two attributes of obia.handlers.geotif.Image that the path reads."""

    def __init__(self, img_data, affine_transformation):
        self.img_data = img_data
        self.affine_transformation = affine_transformation       # [a, b, d, e, xoff, yoff], shapely order


def main():
    rs = np.random.RandomState(0)
    H, W, C = 600, 800, 4
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    data = np.stack([400 * np.sin(xx / (11 + 3 * c)) * np.cos(yy / (13 + 2 * c)) + 1000 + 50 * c + rs.normal(0, 20, (H, W))
                     for c in range(C)], -1).astype(np.float32)
    image = Image(data, [0.5, 0.0, 0.0, -0.5, 300000.0, 2200000.0])           # 0.5 m pixels, north up

    # 1. one raster: SLIC + connectivity + statistics (mean / variance / min / max / skewness / kurtosis per band)
    seg = segment(image, segmentation_bands=[0, 1, 2, 3], statistics_bands=[0, 1, 2, 3], method="slic",
                  n_segments=1500, compactness=0.25)
    print(f"segment(): {int(seg._segments.max())} segments, objects table {seg.segments.shape}")
    seg.segments.to_csv(os.path.join(os.path.dirname(__file__), "quickstart_objects.csv"), index=False)

    # 2. polygons in map coordinates, ids 1..N like the reference's GeoDataFrame
    polys = seg.polygons(image.affine_transformation)
    with open(os.path.join(os.path.dirname(__file__), "quickstart_segments.geojson"), "w") as f:
        json.dump({"type": "FeatureCollection", "features": polys.geojson_features()}, f)
    print(f"polygons(): {len(polys)} polygons, {len(polys.xy)} vertices")

    # 3. the tiled driver (checkerboard tiles with overlap), then texture columns for one band
    mask = np.ones((H, W), np.uint8)
    mask[:40, :60] = 0
    labels, n = create_tiled_segments(data, input_mask=mask, tile_size=256, buffer=32, crown_radius=5, pixel_size=(0.5, 0.5),
                                      compactness=0.25)
    table = create_objects(labels, image, spectral_bands=[0], textural_bands=[0], calculate_textural=True)
    print(f"create_tiled_segments(): {n} segments; create_objects(): columns {list(table.columns)[:5]} ...")

    # 4. consumers of the label raster
    edges = slic_edge(labels)
    pts = [(300000.0 + 0.5 * 400.5, 2200000.0 - 0.5 * 300.5), (300000.0 + 0.5 * 10.5, 2200000.0 - 0.5 * 10.5)]
    labelled, mixed = label_segments(labels, image.affine_transformation, pts, ["canopy", "masked corner"])
    print(f"slic_edge(): {100 * float(edges.mean()):.1f} % edge pixels; label_segments(): {labelled}, mixed {mixed}")
    assert len(polygonize(labels, start_label=1)) == n


if __name__ == "__main__":
    main()
