"""PCIe-inclusive rate of the bench workload: host NumPy raster in, host labels + statistics out (obia_tiled_slic_f32 and
obia_zonal_stats_f32 with host pointers).  Never the bench's `value`; recorded in DESIGN.md 4."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from obia_amd.tiling import create_tiled_segments
from obia_amd.statistics import zonal_stats
H = W = int(os.environ.get("SIZE", 16384))
img = bench.synth_raster(H, W, 8, seed=0, device=torch.device("cuda", 0), row0=0).cpu().numpy()
mask = np.ones((H, W), np.uint8)
torch.cuda.empty_cache()
for rep in range(2):
    t0 = time.time()
    lab, n = create_tiled_segments(img, input_mask=mask, tile_size=2048, buffer=64, crown_radius=5, pixel_size=(0.5, 0.5), compactness=10.0)
    t1 = time.time()
    st = zonal_stats(img, lab, n_labels=n)
    t2 = time.time()
    print(f"host in / host out: tiled SLIC {t1-t0:.3f} s, zonal {t2-t1:.3f} s -> {H*W/(t2-t0)/1e6:.0f} Mpixel/s ({n} segments)", flush=True)
