"""Diagnostic (tools/build_variant.sh stampN -DOBIA_STAMP -DASSIGN_WAVES=N, then OBIA_HIP_LIB=...): per-phase wave-cycle shares of
the sweep kernel on the bench-like masked tiled workload (one 4096^2 raster, tile 2048)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from obia_amd import _lib
from obia_amd.segmentation import slic
lib = _lib.load()
H = W = 4096; C = 8
g = torch.Generator(device="cuda").manual_seed(0)
yy = torch.arange(H, device="cuda", dtype=torch.float32)[:, None]
xx = torch.arange(W, device="cuda", dtype=torch.float32)[None, :]
img = torch.empty((H, W, C), device="cuda", dtype=torch.float32)
for c in range(C):
    img[:, :, c] = 400.0 * torch.sin(xx / (11 + 3 * c)) * torch.cos(yy / (13 + 2 * c)) + 1000 + 50 * c + 20.0 * torch.randn((H, W), device="cuda", generator=g)
n = round(H * W / 324.0)
out = (ctypes.c_ulonglong * 16)()
slic(img, n_segments=n, compactness=10.0, _normalize_bands=True)
torch.cuda.synchronize()
ctypes.CDLL(_lib.LIB_PATH).obia_debug_stamps(out, 1)
slic(img, n_segments=n, compactness=10.0, _normalize_bands=True)
torch.cuda.synchronize()
ctypes.CDLL(_lib.LIB_PATH).obia_debug_stamps(out, 1)
names = ["staging", "load issue", "scoring", "visits", "labels", "run merge", "fold", "barrier+flush"]
tot = sum(out[i] for i in range(8))
print("waves", out[15], "cycles/wave", tot / max(1, out[15]))
for i, nme in enumerate(names):
    print(f"  {nme:14s} {out[i]/max(1,out[15]):10.0f} cyc/wave  {100.0*out[i]/tot:5.1f} %")
cn = ["footprints", "visits", "visits evaluating colours", "j-slices evaluated", "candidate pixels (sum over visits)"]
fp = max(1, out[8])
for i, nme in enumerate(cn):
    print(f"  {nme:36s} {out[8+i]:14d}  per footprint {out[8+i]/fp:8.2f}")
