"""Quick timing probe of the single-raster path (not the bench): prints per-stage HIP-event times."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from obia_amd import _lib
from obia_amd.segmentation import slic
from obia_amd.statistics import zonal_stats


def synth(H, W, C, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    yy = torch.arange(H, device="cuda", dtype=torch.float32)[:, None]
    xx = torch.arange(W, device="cuda", dtype=torch.float32)[None, :]
    out = torch.empty((H, W, C), device="cuda", dtype=torch.float32)
    for c in range(C):
        out[:, :, c] = 400.0 * torch.sin(xx / (11 + 3 * c)) * torch.cos(yy / (13 + 2 * c)) + 1000 + 50 * c \
            + 20.0 * torch.randn((H, W), device="cuda", generator=g)
    return out


ctx = _lib.default_context(0)
ctx.set_profiling(True)
for (H, W, C, n, comp) in [(4096, 4096, 4, 50000, 10.0), (4096, 4096, 8, 50000, 10.0), (8192, 8192, 8, 207000, 10.0),
                           (4096, 4096, 8, 50000, 0.25)]:
    img = synth(H, W, C)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.time()
        lab = slic(img, n_segments=n, compactness=comp, _normalize_bands=True, exit_on_fixed_point=bool(int(os.environ.get('OBIA_FP', '0'))))
        torch.cuda.synchronize()
        dt = time.time() - t0
        t = ctx.timing()
    nl = int(lab.max().item())
    t0 = time.time()
    st = zonal_stats(img, lab, n_labels=nl)
    torch.cuda.synchronize()
    dz = time.time() - t0
    tz = ctx.timing()
    mp = H * W / 1e6
    sweep = t["assign_ms"] / max(1, t["sweeps"])
    CP = (C + 3) // 4 * 4
    print(f"{H}x{W}x{C} c={comp}: wall {dt*1e3:.1f} ms ({mp/dt:.0f} Mpx/s) feat {t['features_ms']:.2f} sweeps {t['assign_ms']:.2f} "
          f"({sweep:.3f} ms/sweep = {mp*1e6*(4*CP+4)/sweep/1e9:.0f} GB/s) cc {t['connectivity_ms']:.2f} total {t['total_ms']:.2f} "
          f"| zonal {tz['zonal_ms']:.2f} ms wall {dz*1e3:.1f} | labels {nl} ws {ctx.workspace_bytes()/2**30:.2f} GiB", flush=True)
    del img, lab
