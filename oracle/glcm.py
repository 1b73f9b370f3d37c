"""CPU restatement of the per-segment GLCM texture statistics (TEST INFRASTRUCTURE ONLY).

Reference: calculate_textural_stats (obia/segmentation/segment_statistics.py:179-298) on the masked bounding-box crop that
create_objects hands over (:478-479, utils/utils.py:37-67).  The reference indexes the crop as ``image[:, :, band]``
although the crop is laid out (bands, h, w) (:214) -- as written it reads one COLUMN of every band.  This restatement
and the HIP kernel implement what the code evidently means, the band plane ``image[band, :, :]``:
  * crop of the band to the segment's bounding box; pixels outside the segment (and NaN pixels) become 0 (:247-248);
  * float data: min / max over that crop INCLUDING the zeros, ``uint8((v - min) / (max - min) * 255)`` in the dtype of
    the data (float32), truncated; constant crop -> all zeros (:253-258); no valid pixel -> every statistic NaN (:217-231);
  * grey-level co-occurrence matrix, distance 2, angles 0, pi/4, pi/2, 3pi/4 (offsets (0,2), (1,1), (2,0), (1,-1) in
    (row, col), rounded like scikit-image does), 256 levels, symmetric, normed (:260-267);
  * contrast, dissimilarity, homogeneity, ASM, energy, correlation per angle (scikit-image greycoprops), then the mean
    over the four angles (:285-296).
The GLCM arithmetic is scikit-image's (third-party, absent from /root/reference); this file restates it and is pinned by
tests/golden/glcm_*.npz, produced by scikit-image 0.18.3's greycomatrix / greycoprops (tests/golden/gen_goldens_glcm.py).
"""
import numpy as np

OFFSETS = [(0, 2), (1, 1), (2, 0), (1, -1)]     # (d_row, d_col) for angles 0, pi/4, pi/2, 3pi/4 at distance 2
PROPS = ("contrast", "dissimilarity", "homogeneity", "ASM", "energy", "correlation")


def quantise_crop(band, inside):
    """band: (h, w) float32 crop; inside: (h, w) bool (pixel belongs to the segment and is not NaN) -> uint8 crop or
    None when no pixel is valid."""
    valid = inside & ~np.isnan(band)
    if not valid.any():
        return None
    clean = np.where(valid, band, np.float32(0)).astype(np.float32)
    lo, hi = clean.min(), clean.max()
    if hi == lo:
        return np.zeros(clean.shape, np.uint8)
    return ((clean - lo) / (hi - lo) * np.float32(255)).astype(np.uint8)


def glcm_props(q):
    """q: (h, w) uint8 -> dict of the six statistics, each the mean over the four angles."""
    h, w = q.shape
    lev = np.arange(256, dtype=np.float64)
    acc = {p: 0.0 for p in PROPS}
    for dr, dc in OFFSETS:
        r0, r1 = max(0, -dr), min(h, h - dr)
        c0, c1 = max(0, -dc), min(w, w - dc)
        P = np.zeros((256, 256), np.float64)
        if r1 > r0 and c1 > c0:
            a = q[r0:r1, c0:c1].ravel()
            b = q[r0 + dr:r1 + dr, c0 + dc:c1 + dc].ravel()
            np.add.at(P, (a, b), 1.0)
            P = P + P.T
        s = P.sum()
        if s > 0:
            P /= s
        I, J = lev[:, None], lev[None, :]
        D = I - J
        acc["contrast"] += (P * D * D).sum()
        acc["dissimilarity"] += (P * np.abs(D)).sum()
        acc["homogeneity"] += (P / (1.0 + D * D)).sum()
        asm = (P * P).sum()
        acc["ASM"] += asm
        acc["energy"] += np.sqrt(asm)
        di, dj = I - (I * P).sum(), J - (J * P).sum()
        si, sj = np.sqrt((P * di * di).sum()), np.sqrt((P * dj * dj).sum())
        acc["correlation"] += 1.0 if (si < 1e-15 or sj < 1e-15) else (P * di * dj).sum() / (si * sj)
    return {p: v / len(OFFSETS) for p, v in acc.items()}


def texture_stats(raw, labels, bands=None, start_label=1, n_labels=None):
    """-> dict prop -> (n_labels, n_bands) float64, NaN where the segment is empty or has no valid pixel in the band."""
    raw = np.asarray(raw)
    lab = np.asarray(labels)
    H, W, C = raw.shape
    bands = list(range(C)) if bands is None else list(bands)
    if n_labels is None:
        n_labels = int(lab.max()) - start_label + 1
    out = {p: np.full((n_labels, len(bands)), np.nan, np.float64) for p in PROPS}
    for i in range(n_labels):
        ys, xs = np.nonzero(lab == start_label + i)
        if ys.size == 0:
            continue
        y0, y1, x0, x1 = ys.min(), ys.max() + 1, xs.min(), xs.max() + 1
        inside = lab[y0:y1, x0:x1] == start_label + i
        for j, b in enumerate(bands):
            q = quantise_crop(raw[y0:y1, x0:x1, b].astype(np.float32), inside)
            if q is None:
                continue
            for p, v in glcm_props(q).items():
                out[p][i, j] = v
    return out
