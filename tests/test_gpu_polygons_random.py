"""Randomised parity of the label raster -> polygon rings pass (f1) against oracle/polygons.py: seeded label maps (blobs, noise,
noisy grids, stripes, nested rings) with unlabelled pixels, sizes from 1 x 1 up, start_label 0 / 1.  Bar: the ring tables are equal
ring for ring (label, hole flag, vertex list) in the table's order."""
import os

import numpy as np
import pytest

from tests.test_gpu_connectivity_random import make_case as make_label_map
from tests.test_gpu_polygons import oracle_grouped, rings_as_tuples

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def make_case(seed):
    rs = np.random.RandomState(13000 + seed)
    lab, _, _ = make_label_map(seed)
    lab = lab.astype(np.int64)
    if seed % 5 == 0:      # nested rings: holes inside holes
        H, W = lab.shape
        yy, xx = np.mgrid[0:H, 0:W]
        r = np.maximum(np.abs(yy - H // 2), np.abs(xx - W // 2))
        lab = (r // int(rs.randint(1, 4))) % 3 + 1
    if seed % 7 == 0:      # tiny maps
        lab = lab[:int(rs.randint(1, 4)), :int(rs.randint(1, 6))]
    lab = lab % 50         # few labels, many rings per label
    start_label = int(rs.choice([0, 1]))
    if start_label == 0:
        lab = lab - 1      # unlabelled pixels: -1
    return np.ascontiguousarray(lab.astype(np.int32)), start_label


@pytest.mark.parametrize("seed", range(int(os.environ.get("OBIA_RANDOM_POLYGON_CASES", "20"))))
def test_random_rings_vs_oracle(seed):
    from obia_amd.polygons import polygonize
    lab, start_label = make_case(seed)
    tab = polygonize(lab, start_label=start_label)
    assert rings_as_tuples(tab) == oracle_grouped(lab, start_label), f"seed {seed}: shape {lab.shape}, start_label {start_label}"
