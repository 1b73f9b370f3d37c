#!/opt/conda/bin/python3.9
"""Golden vectors for the GLCM texture statistics from scikit-image 0.18.3.

Run ONLY in the build container:   /opt/conda/bin/python3.9 tests/golden/gen_goldens_glcm.py
It imports scikit-image's greycomatrix / greycoprops (the functions obia calls, under their newer names, at
obia/segmentation/segment_statistics.py:260-296) and writes one small .npz fixture: inputs (raster as uint16 digital
numbers + NaN mask, label map) and, per (label, band), the six statistics scikit-image produced on the quantised
bounding-box crop of the segment (crop / zero fill / uint8 rescale as oracle/glcm.py documents: the band PLANE, not the
reference's mis-indexed column).  Data only; nothing of scikit-image or the reference is copied.
"""
import os
import warnings

import numpy as np

warnings.filterwarnings("ignore")
from skimage.feature import greycomatrix, greycoprops  # noqa: E402
import skimage  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
PROPS = ("contrast", "dissimilarity", "homogeneity", "ASM", "energy", "correlation")


def main():
    rs = np.random.RandomState(21)
    H, W, C = 90, 110, 3
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    dn = np.empty((H, W, C), np.uint16)
    for c in range(C):
        b = 400.0 * np.sin(xx / (7 + 3 * c)) * np.cos(yy / (5 + 2 * c)) + 1000 + 50 * c + rs.normal(0, 40, (H, W))
        dn[:, :, c] = np.clip(np.rint(b), 0, 65535).astype(np.uint16)
    nanmask = rs.rand(H, W, C) < 0.01
    # irregular segments: blocks with jagged borders, a one-pixel segment, a 2-pixel-high strip, a constant segment
    lab = ((yy // 14) * 9 + (xx + 3 * np.sin(yy / 3.0)) // 13 + 1).astype(np.int32)
    lab[0, 0] = lab.max() + 1
    lab[40:42, 30:70] = lab.max() + 1
    const_label = int(lab[20, 20])
    dn[lab == const_label] = 777
    nanmask[lab == const_label] = False
    allnan_label = int(lab[60, 60])
    nanmask[lab == allnan_label, 1] = True
    raw = dn.astype(np.float32)
    raw[nanmask] = np.nan
    n_labels = int(lab.max())
    out = {p: np.full((n_labels, C), np.nan, np.float64) for p in PROPS}
    for l in range(1, n_labels + 1):
        ys, xs = np.nonzero(lab == l)
        if ys.size == 0:
            continue
        y0, y1, x0, x1 = ys.min(), ys.max() + 1, xs.min(), xs.max() + 1
        inside = lab[y0:y1, x0:x1] == l
        for c in range(C):
            band = raw[y0:y1, x0:x1, c]
            valid = inside & ~np.isnan(band)
            if not valid.any():
                continue
            clean = np.where(valid, band, np.float32(0)).astype(np.float32)
            lo, hi = clean.min(), clean.max()
            q = np.zeros(clean.shape, np.uint8) if hi == lo else ((clean - lo) / (hi - lo) * np.float32(255)).astype(np.uint8)
            glcm = greycomatrix(q, distances=[2], angles=[0, np.pi / 4, np.pi / 2, 3 * np.pi / 4], levels=256,
                                symmetric=True, normed=True)
            for p in PROPS:
                out[p][l - 1, c] = np.mean(greycoprops(glcm, p))
    np.savez_compressed(os.path.join(HERE, "glcm_90x110x3.npz"), dn=dn, nanmask=np.packbits(nanmask), labels=lab,
                        skimage_version=skimage.__version__, **out)
    print("wrote glcm_90x110x3.npz", skimage.__version__, "labels", n_labels)


if __name__ == "__main__":
    main()
