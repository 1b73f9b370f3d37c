import os, sys, ast
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from obia_amd.tiling import create_tiled_segments
from obia_amd.segmentation import slic
z = np.load("tests/golden/quickstart_128x128x3.npz")
raw = z["raw"].astype(np.float32)
kw = dict(tile_size=100, buffer=16, crown_radius=4, pixel_size=(1.0, 1.0), compactness=8)
la, na = create_tiled_segments(raw, **kw)
lb, nb = create_tiled_segments(raw, exit_on_fixed_point=True, **kw)
print(na, nb, (la != lb).sum())
d = np.argwhere((la == 0) != (lb == 0))
print("zero-mismatch", len(d), d[:5])
# partition differences
from tests.metrics import adjusted_rand_index
print("ARI", adjusted_rand_index(la, lb))
ys, xs = np.nonzero(la != lb)
if len(ys):
    print("bbox of differing ids", ys.min(), ys.max(), xs.min(), xs.max())
# per-tile direct slic with masks all ones to isolate: black tile (1,1)
for (y0, y1, x0, x1) in [(0, 100, 0, 100), (100, 128, 100, 128), (0, 116, 84, 128), (84, 128, 0, 116)]:
    t = raw[y0:y1, x0:x1]
    m = np.ones(t.shape[:2], np.uint8)
    n = round(m.sum() / (np.pi * 16))
    a = slic(t, n_segments=n, compactness=8, mask=m, _normalize_bands=True)
    b = slic(t, n_segments=n, compactness=8, mask=m, _normalize_bands=True, exit_on_fixed_point=True)
    print((y0, y1, x0, x1), n, np.array_equal(a, b), len(np.unique(a)), len(np.unique(b)))
