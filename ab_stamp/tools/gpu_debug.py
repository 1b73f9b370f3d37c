"""Diagnostic dump used while bringing kernels up on the GPU box (not a test)."""
import ast, glob, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.metrics import adjusted_rand_index, label_disagreement
from obia_amd.segmentation import slic
from oracle import oracle as orc

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
for p in sorted(glob.glob(os.path.join(GOLD, "*.npz"))):
    name = os.path.basename(p)[:-4]
    if name.startswith(("connectivity_", "quickshift_", "sliczero")):
        continue
    z = np.load(p)
    params = ast.literal_eval(str(z["params"]))
    kw = dict(n_segments=params["n_segments"], compactness=params["compactness"], max_num_iter=params.get("max_iter", 10),
              convert2lab=params.get("convert2lab", None), start_label=params.get("start_label", 1))
    raw = z["raw"].astype(np.float32)
    mask = z["mask"] if "mask" in z.files else None
    try:
        t0 = time.time()
        pre = slic(torch.as_tensor(raw).cuda(), mask=None if mask is None else torch.as_tensor(mask).cuda(),
                   _normalize_bands=True, _stage="pre", **kw).cpu().numpy()
        fin = slic(raw, mask=mask, _normalize_bands=True, min_size_factor=params.get("min_size_factor", 0.5),
                   max_size_factor=params.get("max_size_factor", 3), **kw)
        dt = time.time() - t0
        if mask is None:
            print(f"{name:28s} pre-diff {label_disagreement(pre, z['labels_pre']):.2e}  final ARI {adjusted_rand_index(fin, z['labels']):.5f} "
                  f"exact {np.array_equal(fin, z['labels'])} n {len(np.unique(fin))}/{len(np.unique(z['labels']))} {dt*1e3:.0f} ms", flush=True)
        else:
            o = orc.slic(orc.normalize(raw), mask=mask, n_segments=kw["n_segments"], compactness=kw["compactness"])
            print(f"{name:28s} masked: ARI vs oracle {adjusted_rand_index(fin, o):.5f} exact {np.array_equal(fin, o)} n {len(np.unique(fin))} "
                  f"gold n {len(np.unique(z['labels']))}", flush=True)
    except Exception as e:
        print(f"{name:28s} ERROR {type(e).__name__}: {e}", flush=True)
