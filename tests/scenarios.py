"""Known-answer block-image scenarios, shared by the oracle tests (CPU) and the HIP parity tests.

They re-state, as data, the cases scikit-image's own test-suite pins at the slic()/quickshift()
boundary (skimage/segmentation/tests/test_slic.py: test_color_2d, test_multichannel_2d,
test_gray_2d, test_enforce_connectivity, test_slic_zero, test_more_segments_than_pixels,
test_color_2d_mask, test_multichannel_2d_mask -- SURVEY.md 4).  Each scenario is
(image float32 (H,W,C), kwargs, expected) where expected maps a region (slices) to a label.
"""
import numpy as np


def _noisy(img, scale, seed=0):
    rnd = np.random.RandomState(seed)
    img = img + scale * rnd.normal(size=img.shape)
    return np.clip(img, 0, 1).astype(np.float32)


def color_blocks():
    img = np.zeros((20, 21, 3))
    img[:10, :10, 0] = 1
    img[10:, :10, 1] = 1
    img[10:, 10:, 2] = 1
    return _noisy(img, 0.01)


def multichannel_blocks():
    img = np.zeros((20, 20, 8))
    img[:10, :10, 0:2] = 1
    img[:10, 10:, 2:4] = 1
    img[10:, :10, 4:6] = 1
    img[10:, 10:, 6:8] = 1
    return _noisy(img, 0.01)


def gray_blocks():
    img = np.zeros((20, 21))
    img[:10, :10] = 0.33
    img[10:, :10] = 0.67
    img[10:, 10:] = 1.00
    return _noisy(img, 0.0033)[..., None]


Q = dict(tl=(slice(0, 10), slice(0, 10)), bl=(slice(10, None), slice(0, 10)),
         tr=(slice(0, 10), slice(10, None)), br=(slice(10, None), slice(10, None)))


def slic_scenarios():
    inner = dict(tl=(slice(2, 10), slice(2, 10)), bl=(slice(10, -2), slice(2, 10)),
                 tr=(slice(2, 10), slice(10, -2)), br=(slice(10, -2), slice(10, -2)))
    border = [(slice(0, 2), slice(None)), (slice(-2, None), slice(None)),
              (slice(None), slice(0, 2)), (slice(None), slice(-2, None))]
    out = []
    out.append(("color_2d", color_blocks(), dict(n_segments=4, enforce_connectivity=False, start_label=0),
                [(Q["tl"], 0), (Q["bl"], 2), (Q["tr"], 1), (Q["br"], 3)], 4))
    out.append(("multichannel_2d", multichannel_blocks(), dict(n_segments=4, enforce_connectivity=False, start_label=0),
                [(Q["tl"], 0), (Q["bl"], 2), (Q["tr"], 1), (Q["br"], 3)], 4))
    out.append(("gray_2d", gray_blocks(), dict(n_segments=4, compactness=1, convert2lab=False, start_label=0),
                [(Q["tl"], 0), (Q["bl"], 2), (Q["tr"], 1), (Q["br"], 3)], 4))
    out.append(("slic_zero", color_blocks(), dict(n_segments=4, slic_zero=True, start_label=0),
                [(Q["tl"], 0), (Q["bl"], 2), (Q["tr"], 1), (Q["br"], 3)], 4))
    msk = np.zeros((20, 21), np.uint8)
    msk[2:-2, 2:-2] = 1
    exp = [(inner[k], None) for k in inner]    # labels are seed-order dependent: only the partition is pinned
    exp += [(b, 0) for b in border]
    out.append(("color_2d_mask", color_blocks(), dict(n_segments=4, enforce_connectivity=False, mask=msk), exp, 5))
    msk2 = np.zeros((20, 20), np.uint8)
    msk2[2:-2, 2:-2] = 1
    exp2 = [(inner[k], None) for k in inner]
    exp2 += [(b, 0) for b in border]
    out.append(("multichannel_2d_mask", multichannel_blocks(), dict(n_segments=4, enforce_connectivity=False, mask=msk2), exp2, 5))
    return out


def check_expected(seg, expected, n_unique):
    assert len(np.unique(seg)) == n_unique
    seen = set()
    for region, lab in expected:
        vals = np.unique(seg[region])
        assert vals.size == 1, f"region {region} not uniform: {vals}"
        if lab is not None:
            assert vals[0] == lab, f"region {region}: {vals[0]} != {lab}"
        else:
            assert vals[0] not in seen and vals[0] != 0
            seen.add(int(vals[0]))


CONNECTIVITY_IMG = np.array([[0, 0, 0, 1, 1, 1],
                             [1, 0, 0, 1, 1, 0],
                             [0, 0, 0, 1, 1, 0]], np.float32)[..., None]
CONNECTIVITY_CONNECTED = np.array([[0, 0, 0, 1, 1, 1],
                                   [0, 0, 0, 1, 1, 1],
                                   [0, 0, 0, 1, 1, 1]])
CONNECTIVITY_DISCONNECTED = np.array([[0, 0, 0, 1, 1, 1],
                                      [1, 0, 0, 1, 1, 0],
                                      [0, 0, 0, 1, 1, 0]])
