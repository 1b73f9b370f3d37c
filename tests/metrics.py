"""Label-map comparison metrics (SURVEY.md 8c definitions), int64/float64, no sklearn."""
import numpy as np


def adjusted_rand_index(a, b, ignore=None):
    a = np.asarray(a).ravel().astype(np.int64)
    b = np.asarray(b).ravel().astype(np.int64)
    if ignore is not None:
        keep = (a != ignore) & (b != ignore)
        a, b = a[keep], b[keep]
    _, ai = np.unique(a, return_inverse=True)
    _, bi = np.unique(b, return_inverse=True)
    na, nb = ai.max() + 1, bi.max() + 1
    cont = np.bincount(ai.astype(np.int64) * nb + bi, minlength=na * nb).astype(np.float64)
    sum_comb = (cont * (cont - 1) / 2).sum()
    ra = np.bincount(ai, minlength=na).astype(np.float64)
    rb = np.bincount(bi, minlength=nb).astype(np.float64)
    sa = (ra * (ra - 1) / 2).sum()
    sb = (rb * (rb - 1) / 2).sum()
    n = float(a.size)
    total = n * (n - 1) / 2
    expected = sa * sb / total
    mx = 0.5 * (sa + sb)
    if mx == expected:
        return 1.0
    return float((sum_comb - expected) / (mx - expected))


def boundary_map(labels):
    """pixel whose right or lower 4-neighbour carries a different label"""
    l = np.asarray(labels)
    b = np.zeros(l.shape, bool)
    b[:, :-1] |= l[:, :-1] != l[:, 1:]
    b[:-1, :] |= l[:-1, :] != l[1:, :]
    return b


def _dilate1(b):
    out = b.copy()
    H, W = b.shape
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            ys = slice(max(0, dy), H + min(0, dy))
            xs = slice(max(0, dx), W + min(0, dx))
            yd = slice(max(0, -dy), H + min(0, -dy))
            xd = slice(max(0, -dx), W + min(0, -dx))
            out[yd, xd] |= b[ys, xs]
    return out


def boundary_recall_precision(ref, test):
    """recall = fraction of ref boundary pixels with a test boundary pixel within Chebyshev 1"""
    br, bt = boundary_map(ref), boundary_map(test)
    rec = (br & _dilate1(bt)).sum() / max(1, br.sum())
    prec = (bt & _dilate1(br)).sum() / max(1, bt.sum())
    return float(rec), float(prec)


def label_disagreement(a, b):
    return float((np.asarray(a) != np.asarray(b)).mean())


def check_connected_consecutive(labels, start_label=1, ignore=None):
    """every label 4-connected and labels consecutive from start_label (property test helper)"""
    from scipy import ndimage
    l = np.asarray(labels)
    vals = np.unique(l[l != ignore]) if ignore is not None else np.unique(l)
    assert vals[0] == start_label and (np.diff(vals) == 1).all(), "labels not consecutive"
    # count 4-connected components of the equal-label relation in one pass
    H, W = l.shape
    key = l.astype(np.int64)
    ncomp = 0
    structure = ndimage.generate_binary_structure(2, 1)
    # components of each label == components of the partition; use label on boundaries-free trick
    for v in vals:
        _, n = ndimage.label(key == v, structure)
        ncomp += n
    assert ncomp == len(vals), f"{ncomp} components for {len(vals)} labels"
