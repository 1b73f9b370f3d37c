"""Developer probe: the skewness / kurtosis pass (zonal_moments_kernel) alone on the headline raster."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from obia_amd import _lib
from obia_amd.statistics import zonal_stats
from obia_amd.tiling import create_tiled_segments
from bench import synth_raster
S = int(os.environ.get("ZT_SIZE", "16384")); C = int(os.environ.get("ZT_BANDS", "8"))
dev = torch.device("cuda:0")
img = synth_raster(S, S, C, seed=0, device=dev)
mask = torch.ones((S, S), dtype=torch.uint8, device=dev)
ctx = _lib.default_context(0)
lab, n = create_tiled_segments(img, input_mask=mask, tile_size=2048, buffer=64, crown_radius=5, pixel_size=(0.5, 0.5), compactness=10.0, ctx=ctx)
ts = []
for rep in range(6):
    torch.cuda.synchronize(); t0 = time.time()
    st = zonal_stats(img, lab, n_labels=n, ctx=ctx, moments=True)
    torch.cuda.synchronize(); ts.append((time.time() - t0) * 1e3)
ts = sorted(ts[1:])
print(f"{os.environ.get('OBIA_HIP_LIB', 'default')}: stats + moments wall {ts[len(ts)//2]:.3f} ms (min {ts[0]:.3f}); skew checksum {float(torch.nan_to_num(st['skewness']).sum()):.6e} kurt {float(torch.nan_to_num(st['kurtosis']).sum()):.6e}", flush=True)
