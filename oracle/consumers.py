"""CPU restatements for the consumers of the label raster (TEST INFRASTRUCTURE ONLY).

edge_raster / percentile_stretch follow obia/utils/cost.py:44-48 (`slic_edge`) and :21-26 (`normalise`); the
reference's functions cannot be imported here (rasterio / geopandas are not installed), so these are pinned only by
hand-made known answers (tests/test_oracle_known_answers.py): parity unpinned otherwise.
"""
import numpy as np


def percentile_stretch(a, q_lo=2.0, q_hi=98.0):
    """Clip to the q_lo / q_hi percentiles (NaN ignored), rescale to [0, 1]; undefined results (0/0) become 0."""
    a = np.asarray(a)
    p_lo, p_hi = np.nanpercentile(a, [q_lo, q_hi])
    with np.errstate(all="ignore"):
        scaled = (np.minimum(np.maximum(a, p_lo), p_hi) - p_lo) / (p_hi - p_lo)
    scaled = np.array(scaled, copy=True)
    scaled[np.isnan(scaled)] = 0
    return scaled


def edge_raster(lab):
    """1.0 where a pixel's label differs from the pixel below it or from the pixel to its right, stretched like every
    layer of the cost surface."""
    lab = np.asarray(lab)
    differs_down = np.zeros(lab.shape, bool)
    differs_right = np.zeros(lab.shape, bool)
    differs_down[:-1] = lab[1:] != lab[:-1]
    differs_right[:, :-1] = lab[:, 1:] != lab[:, :-1]
    return percentile_stretch((differs_down | differs_right).astype(np.float32))
