"""CPU restatement of the tile loops of obia.utils.tiling.create_tiled_segments on label rasters.
TEST INFRASTRUCTURE ONLY (see oracle/obia_oracle.c header for the rules).

The reference (obia/utils/tiling.py:103-291) keeps GeoDataFrames of polygons and needs GDAL / shapely /
geopandas, none of which exist in the build container: PARITY UNPINNED for this stage -- there is no
executable reference and no fixture of its output.  This file restates the algorithm from the source
text, with the polygon predicates replaced by the equivalent pixel counts (tiling.hip header), and the
per-tile segmentation done by the pinned SLIC oracle with the build's masked-grid seeding rule.

OracleTiler mirrors the session API of the HIP library (obia_tiler_* in include/obia_hip.h): a local block of
rows [row0, row0+H) of a (Hg, W) raster, passes by global tile-row range and parity, externally registered
segments.  It is also the compute engine of the CPU (gloo) test of the sharded driver.
"""
import math

import numpy as np

from . import oracle as orc

HUGE = 0xFFFFFFFF


class OracleTiler:
    def __init__(self, img, mask, Hg, row0, tile_size=200, buffer=30, crown_radius=5, pixel_size=(1.0, 1.0),
                 n_segments=None, compactness=10.0, max_iter=10, min_size_factor=0.5, max_size_factor=3, sigma=0, spacing=None):
        self.spacing = spacing
        self.sigma = sigma              # forwarded to slic() like every other keyword (tiling.py:137-143)
        self.img = np.asarray(img, np.float32)
        self.H, self.W, self.C = self.img.shape
        self.Hg, self.row0 = int(Hg), int(row0)
        self.inmask = np.ones((self.H, self.W), bool) if mask is None else (np.asarray(mask) != 0)
        self.T, self.B = int(tile_size), int(buffer)
        self.crown_radius, (self.pw, self.ph) = crown_radius, pixel_size
        self.n_segments, self.compactness, self.max_iter, self.msf = n_segments, compactness, max_iter, min_size_factor
        self.xsf = max_size_factor      # slic's default 3: create_tiled_segments forwards **kwargs untouched (tiling.py:137-143)
        self.G = np.zeros((self.H, self.W), np.int64)
        self.sizes = {}
        self.alive = {}
        self.next_id = 1
        self.untouched_white_tiles = []  # white tiles (tj, ti) that took the else branch of tiling.py:212 (tests)
        # corner squares of side buffer/2 MAP units (tiling.py:189-203) in pixels, per axis -- the side need not be a whole number of
        # pixels, and the reference uses the squares in three ways (pixel k counted from the square's outer edge spans
        # [k * px, (k + 1) * px)):
        #   cl*     centre inside,        (k + 0.5) * px <  cl : what rasterize() burns (all_touched=False, tiling.py:245-255)
        #   cl*_in  wholly inside,        (k + 1)   * px <= cl : a segment of such pixels only has no area in tile_polygon -- neither
        #                                                        `within` nor `overlaps`, it is not selected (tiling.py:205-210)
        #   cl*_any meets the interior,    k        * px <  cl : a segment with such a pixel is not `within` (tiling.py:220-231)
        cl = self.B / 2.0

        def count(px, kind):
            if self.B <= 0:
                return 0
            holds = {0: lambda i: (i + 0.5) * px < cl, 1: lambda i: (i + 1.0) * px <= cl, 2: lambda i: i * px < cl}[kind]
            k = max(0, int(math.floor(cl / px)) + 2)
            while k > 0 and not holds(k - 1):
                k -= 1
            return k
        self.clx, self.cly = count(self.pw, 0), count(self.ph, 0)
        self.clx_in, self.cly_in = count(self.pw, 1), count(self.ph, 1)
        self.clx_any, self.cly_any = count(self.pw, 2), count(self.ph, 2)

    def set_segments(self, first_id, sizes):
        for i, s in enumerate(sizes):
            self.sizes[first_id + i] = int(s)
            self.alive[first_id + i] = True
        self.next_id = max(self.next_id, first_id + len(sizes))

    def _run_tile(self, y0, x0, h, w, tmask):
        tile = self.img[y0:y0 + h, x0:x0 + w].copy()
        if any(tile[:, :, c].max() == tile[:, :, c].min() for c in range(self.C)) or not np.isfinite(tile).all():
            return                                            # NaN features -> ValueError -> "empty tile"
        nvalid = int(tmask.sum())
        if self.n_segments is not None:
            n = round(self.n_segments * nvalid / float(self.T * self.T))
        else:
            n = round(nvalid * self.pw * self.ph / (math.pi * self.crown_radius ** 2))      # tiling.py:126-135
        if n < 1 or nvalid == 0:
            return
        lab = orc.slic(orc.normalize(tile), n_segments=int(n), compactness=self.compactness, max_iter=self.max_iter,
                       mask=tmask.astype(np.uint8), min_size_factor=self.msf, max_size_factor=self.xsf, sigma=self.sigma, spacing=self.spacing)
        sub = self.G[y0:y0 + h, x0:x0 + w]
        for l in np.unique(lab[lab > 0]):                      # labels are consecutive in first-pixel order
            sel = lab == l
            sub[sel] = self.next_id
            self.sizes[self.next_id] = int(sel.sum())
            self.alive[self.next_id] = True
            self.next_id += 1

    def run(self, white, tr_lo, tr_hi, parity=-1):
        T, B, Hg, W = self.T, self.B, self.Hg, self.W
        nty = -(-Hg // T)
        for tj in range(max(0, tr_lo), min(nty, tr_hi)):
            if parity >= 0 and (tj & 1) != parity:
                continue
            for ti in range(-(-W // T)):
                if ((ti + tj) % 2 != 0) != bool(white):
                    continue
                if not white:                                   # pass 1: black tiles (tiling.py:103-153)
                    gy0, h, x0, w = tj * T, min(T, Hg - tj * T), ti * T, min(T, W - ti * T)
                    y0 = gy0 - self.row0
                    assert 0 <= y0 and y0 + h <= self.H
                    self._run_tile(y0, x0, h, w, self.inmask[y0:y0 + h, x0:x0 + w])
                    continue
                gy0, gy1 = max(0, tj * T - B), min(Hg, tj * T + T + B)      # pass 2: white tiles (tiling.py:156-287)
                x0, x1 = max(0, ti * T - B), min(W, ti * T + T + B)
                y0, y1 = gy0 - self.row0, gy1 - self.row0
                assert 0 <= y0 and y1 <= self.H, "halo too small"
                h, w = y1 - y0, x1 - x0
                def squares(cy, cx):
                    m = np.zeros((h, w), bool)
                    cy, cx = min(cy, h), min(cx, w)
                    if cy > 0 and cx > 0:
                        m[h - cy:, :cx] = True
                        m[h - cy:, w - cx:] = True
                    return m
                corner = squares(self.cly, self.clx)            # burned into the mask
                corner_in = squares(self.cly_in, self.clx_in)   # pixels with no area in tile_polygon
                corner_any = squares(self.cly_any, self.clx_any)  # pixels the squares reach at all
                sub = self.G[y0:y1, x0:x1]
                tmask = self.inmask[y0:y1, x0:x1].copy()
                area = sub[~corner_in]                          # tile_polygon = window minus the corner squares (:187-203)
                ids = np.unique(area[area > 0])                 # segments with area in the polygon: within or overlaps (:205-210)
                clear = sub[~corner_any]
                if len(ids):                                   # some segment is within / overlaps the polygon (:205-212)
                    for g in ids:
                        if int((clear == g).sum()) == self.sizes[g]:   # within(tile_polygon): dropped (:220-231)
                            sub[sub == g] = 0
                            self.alive[g] = False
                    tmask[sub > 0] = False                     # overlaps: kept and masked out (:213-218, :233-244, :257-258)
                    tmask[corner] = False                      # the corner squares join the mask in THIS branch only (:245-246)
                else:
                    self.untouched_white_tiles.append((tj, ti))
                # else (:261-262): no segment intersects the polygon -- the mask is left as read, the corner squares
                # ARE segmented (and a segment that lies wholly inside a corner square is written over in G: a label
                # raster cannot hold the reference's two overlapping polygons)
                self._run_tile(y0, x0, h, w, tmask)

    def finalize(self):
        order = [g for g in range(1, self.next_id) if self.alive.get(g)]
        lut = np.zeros(self.next_id + 1, np.int64)
        lut[order] = np.arange(1, len(order) + 1)              # segment_id = 1..N (tiling.py:289-290)
        return lut[self.G], len(order)


def create_tiled_segments(img, mask=None, tile_size=200, buffer=30, crown_radius=5, pixel_size=(1.0, 1.0),
                          n_segments=None, compactness=10.0, max_iter=10, min_size_factor=0.5, max_size_factor=3, white_order=0, sigma=0, spacing=None):
    """white_order 0: the reference's raster order of white tiles; 1: even tile rows, then odd tile rows
    (the order of the sharded driver)."""
    img = np.asarray(img, np.float32)
    H = img.shape[0]
    t = OracleTiler(img, mask, H, 0, tile_size, buffer, crown_radius, pixel_size, n_segments, compactness, max_iter,
                    min_size_factor, max_size_factor, sigma, spacing)
    nty = -(-H // tile_size)
    t.run(False, 0, nty)
    if white_order == 1:
        t.run(True, 0, nty, 0)
        t.run(True, 0, nty, 1)
    else:
        t.run(True, 0, nty)
    return t.finalize()
