"""CPU restatement of the tile loops of obia.utils.tiling.create_tiled_segments on label rasters.
TEST INFRASTRUCTURE ONLY (see oracle/obia_oracle.c header for the rules).

The reference (obia/utils/tiling.py:103-291) keeps GeoDataFrames of polygons and needs GDAL / shapely /
geopandas, none of which exist in the build container: PARITY UNPINNED for this stage -- there is no
executable reference and no fixture of its output.  This file restates the algorithm from the source
text, with the polygon predicates replaced by the equivalent pixel counts (tiling.hip header), and the
per-tile segmentation done by the pinned SLIC oracle with the build's masked-grid seeding rule.
"""
import math

import numpy as np

from . import oracle as orc


def create_tiled_segments(img, mask=None, tile_size=200, buffer=30, crown_radius=5, pixel_size=(1.0, 1.0),
                          n_segments=None, compactness=10.0, max_iter=10, min_size_factor=0.5):
    img = np.asarray(img, np.float32)
    H, W, C = img.shape
    inmask = np.ones((H, W), bool) if mask is None else (np.asarray(mask) != 0)
    pw, ph = pixel_size
    G = np.zeros((H, W), np.int64)
    sizes = {}
    alive = {}
    next_id = 1
    cl = buffer / 2.0
    clx = max(0, int(math.ceil(cl / pw - 0.5))) if buffer > 0 else 0
    cly = max(0, int(math.ceil(cl / ph - 0.5))) if buffer > 0 else 0

    def run_tile(y0, x0, h, w, tmask):
        nonlocal next_id
        tile = img[y0:y0 + h, x0:x0 + w].copy()
        if any(tile[:, :, c].max() == tile[:, :, c].min() for c in range(C)) or not np.isfinite(tile).all():
            return                                            # NaN features -> ValueError -> "empty tile"
        nvalid = int(tmask.sum())
        if n_segments is not None:
            n = round(n_segments * nvalid / float(tile_size * tile_size))
        else:
            n = round(nvalid * pw * ph / (math.pi * crown_radius ** 2))      # tiling.py:126-135
        if n < 1 or nvalid == 0:
            return
        lab = orc.slic(orc.normalize(tile), n_segments=int(n), compactness=compactness, max_iter=max_iter,
                       mask=tmask.astype(np.uint8), min_size_factor=min_size_factor, max_size_factor=1e9)
        ids = np.unique(lab[lab > 0])
        sub = G[y0:y0 + h, x0:x0 + w]
        for l in ids:                                          # labels are consecutive in first-pixel order
            sel = lab == l
            sub[sel] = next_id
            sizes[next_id] = int(sel.sum())
            alive[next_id] = True
            next_id += 1

    T = tile_size
    for j in range(0, H, T):                                   # pass 1: black tiles (tiling.py:103-153)
        for i in range(0, W, T):
            if (i // T + j // T) % 2 != 0:
                continue
            h, w = min(T, H - j), min(T, W - i)
            run_tile(j, i, h, w, inmask[j:j + h, i:i + w])
    for j in range(0, H, T):                                   # pass 2: white tiles (tiling.py:156-287)
        for i in range(0, W, T):
            if (i // T + j // T) % 2 == 0:
                continue
            y0, y1 = max(0, j - buffer), min(H, j + T + buffer)
            x0, x1 = max(0, i - buffer), min(W, i + T + buffer)
            h, w = y1 - y0, x1 - x0
            corner = np.zeros((h, w), bool)
            cy, cx = min(cly, h), min(clx, w)
            if cy > 0 and cx > 0:
                corner[h - cy:, :cx] = True
                corner[h - cy:, w - cx:] = True
            sub = G[y0:y1, x0:x1]
            tmask = inmask[y0:y1, x0:x1].copy()
            inside = sub[~corner]
            ids, cnt = np.unique(inside[inside > 0], return_counts=True)
            for g, c in zip(ids, cnt):
                if c == sizes[g]:                              # within(tile_polygon): dropped (:220-231)
                    sub[sub == g] = 0
                    alive[g] = False
            tmask[sub > 0] = False                             # overlaps: kept and masked out (:213-255)
            tmask[corner] = False                              # corner squares (:189-203, :248)
            run_tile(y0, x0, h, w, tmask)
    order = [g for g in range(1, next_id) if alive.get(g)]
    lut = np.zeros(next_id + 1, np.int64)
    lut[order] = np.arange(1, len(order) + 1)                  # segment_id = 1..N (tiling.py:289-290)
    return lut[G], len(order)
