"""Single-raster SLIC run for counter collection (rocprofv3 --pmc ... -- python3 tools/prof_slic.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from obia_amd.segmentation import slic
H = W = int(os.environ.get("OBIA_PROF_SIZE", "4096"))
C = int(os.environ.get("OBIA_PROF_BANDS", "8"))
g = torch.Generator(device="cuda").manual_seed(0)
yy = torch.arange(H, device="cuda", dtype=torch.float32)[:, None]
xx = torch.arange(W, device="cuda", dtype=torch.float32)[None, :]
img = torch.empty((H, W, C), device="cuda", dtype=torch.float32)
for c in range(C):
    img[:, :, c] = 400.0 * torch.sin(xx / (11 + 3 * c)) * torch.cos(yy / (13 + 2 * c)) + 1000 + 50 * c + 20.0 * torch.randn((H, W), device="cuda", generator=g)
n = round(H * W / 324.0)
for _ in range(2):
    lab = slic(img, n_segments=n, compactness=10.0, _normalize_bands=True)
torch.cuda.synchronize()
print("labels", int(lab.max().item()))
