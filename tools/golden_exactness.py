"""Developer tool: how many pixels of every scikit-image golden differ from the HIP path, before and after connectivity
(the numbers behind the allow-list of tests/test_gpu_parity.py).   gpurun -- 'python tools/golden_exactness.py'"""
import ast, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from obia_amd.segmentation import slic
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
skip = ("connectivity_", "quickshift_", "moments_", "glcm_", "sigma", "spacing")
for p in sorted(glob.glob(os.path.join(GOLD, "*.npz"))):
    name = os.path.basename(p)[:-4]
    if name.startswith(skip):
        continue
    z = np.load(p)
    params = ast.literal_eval(str(z["params"]))
    kw = dict(n_segments=params["n_segments"], compactness=params["compactness"], max_num_iter=params.get("max_iter", 10),
              convert2lab=params.get("convert2lab", None), start_label=params.get("start_label", 1))
    if params.get("slic_zero"):
        kw["slic_zero"] = True
    raw = z["raw"].astype(np.float32)
    if name.startswith("mask"):   # maskSLIC pinned on scikit-image's own seeds (the seeds input of the C ABI)
        mask, seeds = z["mask"], (z["seeds_yx"], z["seed_steps_all"])
        kw2 = dict(kw, min_size_factor=params.get("min_size_factor", 0.5), max_size_factor=params.get("max_size_factor", 3))
        dev = torch.as_tensor(raw).cuda()
        pre = slic(dev, mask=mask, seeds=seeds, _normalize_bands=True, _stage="pre", **kw2).cpu().numpy()
        lab = slic(dev, mask=mask, seeds=seeds, _normalize_bands=True, **kw2).cpu().numpy()
        print("%-34s msk pre %6d px  final %6d px  of %d" % (name, int((pre != z["labels_pre"]).sum()), int((lab != z["labels"]).sum()), pre.size))
        continue
    pre = slic(torch.as_tensor(raw).cuda(), enforce_connectivity=False, _normalize_bands=True, _stage="pre", **kw).cpu().numpy()
    lab = slic(raw, _normalize_bands=True, min_size_factor=params.get("min_size_factor", 0.5), max_size_factor=params.get("max_size_factor", 3), **kw)
    lab3 = raw.shape[2] == 3 and params.get("convert2lab", None) is not False
    print("%-34s %s pre %6d px  final %6d px  of %d" % (name, "Lab" if lab3 else "   ", int((pre != z["labels_pre"]).sum()), int((lab != z["labels"]).sum()), pre.size))
