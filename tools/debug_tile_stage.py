"""Developer tool: for one seeded tiler case, replay every tile the ORACLE tiler segments through the single-raster operators (HIP vs oracle):
labels before connectivity, connectivity alone on the oracle's pre-labels, final labels -- names the stage where a tile differs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests.test_gpu_tiling_random import make_case
from oracle import tiler, oracle as orc
from obia_amd.segmentation import slic, enforce_connectivity
seed = int(sys.argv[1])
img, mask, kw = make_case(seed)
calls = []
orig = orc.slic
def spy(image, **k):
    calls.append((np.array(image), dict(k)))
    return orig(image, **k)
orc.slic = spy
ref, n_ref = tiler.create_tiled_segments(img, mask, **kw)
orc.slic = orig
print("tiles segmented by the oracle:", len(calls))
for i, (im, k) in enumerate(calls):
    lab_o, pre_o, cent = orig(im, return_all=True, **k)
    m = k["mask"]
    t = torch.as_tensor(im).cuda()
    kk = dict(n_segments=k["n_segments"], compactness=k["compactness"], max_num_iter=k["max_iter"], mask=m, min_size_factor=k["min_size_factor"], max_size_factor=k["max_size_factor"])
    pre_h = slic(t, enforce_connectivity=False, _stage="pre", **kk).cpu().numpy()
    lab_h = slic(t, **kk).cpu().numpy()
    nv = int(m.sum()); K = len(cent)
    seg = nv / K
    cc_h, _ = enforce_connectivity(torch.as_tensor(pre_o.astype(np.int32)).cuda(), int(k["min_size_factor"] * seg), int(k["max_size_factor"] * seg), start_label=1)
    cc_h = cc_h.cpu().numpy()
    print(f"tile {i} shape {im.shape[:2]} K {K} valid {nv}: pre differ {int((pre_h != pre_o).sum())}, connectivity alone differ {int((cc_h != lab_o).sum())}, final differ {int((lab_h != lab_o).sum())}", flush=True)
    if (cc_h != lab_o).any():
        d = np.argwhere(cc_h != lab_o)
        for (y, x) in d[:8]:
            print("     cc px", int(y), int(x), "hip", int(cc_h[y, x]), "oracle", int(lab_o[y, x]), "pre", int(pre_o[y, x]), "pre 3x3:", pre_o[max(0, y - 1):y + 2, max(0, x - 1):x + 2].tolist())
