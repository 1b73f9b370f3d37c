"""Developer tool: one seeded case of tests/test_gpu_tiling_random.py (argv[1] = seed), HIP tiler vs oracle tiler."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests.test_gpu_tiling_random import make_case
from obia_amd.tiling import create_tiled_segments
from oracle import tiler
seed = int(sys.argv[1])
img, mask, kw = make_case(seed)
print("case", seed, img.shape, kw, None if mask is None else int(mask.sum()), flush=True)
ref, n_ref = tiler.create_tiled_segments(img, mask, **kw)
print("oracle", n_ref, flush=True)
lab, n = create_tiled_segments(torch.as_tensor(img).cuda(), input_mask=mask, **kw)
torch.cuda.synchronize()
lab = lab.cpu().numpy()
print("hip", n, "differing pixels", int((lab != ref).sum()), flush=True)
d = np.argwhere(lab != ref)
for (y, x) in d[:20]:
    print("  px", int(y), int(x), "hip", int(lab[y, x]), "oracle", int(ref[y, x]), "tile", int(y) // kw["tile_size"], int(x) // kw["tile_size"], flush=True)
for name, env in (("grouped prep", {"OBIA_PREP_GROUPED": "1"}), ("no colour bound", {"OBIA_COLOUR_BOUND": "0"}), ("store all labels", {"OBIA_STORE_ALL_LABELS": "1"}),
                  ("white features in line", {"OBIA_WHITE_FEATURES_BESIDE": "0"})):
    import subprocess
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); from tests.test_gpu_tiling_random import make_case; from obia_amd.tiling import create_tiled_segments; "
            "from oracle import tiler; img, mask, kw = make_case(%d); ref, n_ref = tiler.create_tiled_segments(img, mask, **kw); "
            "lab, n = create_tiled_segments(torch.as_tensor(img).cuda(), input_mask=mask, **kw); print(int((lab.cpu().numpy() != ref).sum()))") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), seed)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env))
    print(name, "->", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
