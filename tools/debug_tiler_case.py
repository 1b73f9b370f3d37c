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
