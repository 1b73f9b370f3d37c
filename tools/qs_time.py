"""Timing probe for quickshift (BASELINE config 5: 8192x8192x3, kernel_size=5, max_dist=10)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from obia_amd.segmentation import quickshift
for S in (1024, 2048, 4096, 8192):
    g = torch.Generator(device="cuda").manual_seed(0)
    yy = torch.arange(S, device="cuda", dtype=torch.float32)[:, None]
    xx = torch.arange(S, device="cuda", dtype=torch.float32)[None, :]
    img = torch.empty((S, S, 3), device="cuda", dtype=torch.float32)
    for c in range(3):
        img[:, :, c] = 400.0 * torch.sin(xx / (11 + 3 * c)) * torch.cos(yy / (13 + 2 * c)) + 1000 + 50 * c + 20.0 * torch.randn((S, S), device="cuda", generator=g)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        lab = quickshift(img, ratio=1.0, kernel_size=5, max_dist=10, convert2lab=True, rng=os.environ.get("QS_RNG", 42) if os.environ.get("QS_RNG") != "42" else 42, _normalize_bands=True)
        torch.cuda.synchronize(); dt = time.time() - t0
    print(f"{S}x{S}x3 quickshift ks=5: {dt*1e3:.1f} ms  {S*S/dt/1e6:.1f} Mpixel/s  labels {int(lab.max().item())+1}", flush=True)
