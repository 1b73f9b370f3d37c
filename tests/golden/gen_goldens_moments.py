#!/usr/bin/env python3
"""Golden vectors for the skewness / kurtosis columns from SciPy itself.

Run in the build container with the system interpreter (scipy 1.15.3, NumPy 2.2):
    python3 tests/golden/gen_goldens_moments.py
It imports scipy.stats -- the third-party library that holds the arithmetic obia calls at
obia/segmentation/segment_statistics.py:173-175 (`skew(band_flat)`, `kurtosis(band_flat)`; the reference asks for
scipy>=1.14.1) -- and writes one small .npz fixture: inputs (float32 raster as uint16 digital numbers + NaN mask,
label map) and the per-(label, band) outputs SciPy produced on the float32 pixels of each label, NaN pixels dropped
per band exactly as calculate_spectral_stats does (:145-147).  Data only; nothing of SciPy or the reference is copied.
"""
import os
import warnings

import numpy as np
import scipy
from scipy.stats import kurtosis, skew

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rs = np.random.RandomState(7)
    H, W, C = 96, 131, 5
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    dn = np.empty((H, W, C), np.uint16)
    for c in range(C):
        b = 400.0 * np.sin(xx / (11 + 3 * c)) * np.cos(yy / (13 + 2 * c)) + 1000 + 50 * c + rs.gamma(2.0, 15.0, (H, W))
        dn[:, :, c] = np.clip(np.rint(b), 0, 65535).astype(np.uint16)
    nanmask = rs.rand(H, W, C) < 0.01
    # labels: irregular blocks, one constant-valued segment (label 3) and one all-NaN (label, band)
    lab = ((yy // 13) * 11 + xx // 12 + 1).astype(np.int32)
    lab[rs.rand(H, W) < 0.02] = 0
    dn[lab == 3] = 1234                       # constant data: scipy returns NaN (m2 <= (eps*mean)^2)
    nanmask[lab == 3] = False
    nanmask[lab == 5, 2] = True               # band 2 of label 5 is all NaN
    raw = dn.astype(np.float32)
    raw[nanmask] = np.nan
    n_labels = int(lab.max())
    sk = np.full((n_labels, C), np.nan, np.float64)
    ku = np.full((n_labels, C), np.nan, np.float64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for l in range(1, n_labels + 1):
            sel = lab == l
            for c in range(C):
                v = raw[:, :, c][sel]
                v = v[~np.isnan(v)]
                if v.size == 0:
                    continue
                sk[l - 1, c] = skew(v)
                ku[l - 1, c] = kurtosis(v)
    np.savez_compressed(os.path.join(HERE, "moments_96x131x5.npz"), dn=dn, nanmask=np.packbits(nanmask), labels=lab,
                        skewness=sk, kurtosis=ku, scipy_version=scipy.__version__, numpy_version=np.__version__)
    print("wrote moments_96x131x5.npz", scipy.__version__, "NaN skew entries:", int(np.isnan(sk).sum()))


if __name__ == "__main__":
    main()
