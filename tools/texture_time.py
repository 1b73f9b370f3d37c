"""Time the skewness/kurtosis and GLCM passes on a tiled-SLIC label map (default 8192^2 x 8, ~206 k segments)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from obia_amd.tiling import create_tiled_segments
from obia_amd.statistics import zonal_stats, texture_stats
H = W = int(os.environ.get("SIZE", 8192))
img = bench.synth_raster(H, W, 8, seed=0, device=torch.device("cuda", 0), row0=0)
mask = torch.ones((H, W), dtype=torch.uint8, device="cuda")
lab, n = create_tiled_segments(img, input_mask=mask, tile_size=2048, buffer=64, crown_radius=5, pixel_size=(0.5, 0.5), compactness=10.0)
for name, fn in (("zonal mean/var/min/max", lambda: zonal_stats(img, lab, n_labels=n)),
                 ("zonal + skewness/kurtosis", lambda: zonal_stats(img, lab, n_labels=n, moments=True)),
                 ("GLCM texture, 8 bands", lambda: texture_stats(img, lab, n_labels=n)),
                 ("GLCM texture, 1 band", lambda: texture_stats(img, lab, bands=[0], n_labels=n))):
    fn(); torch.cuda.synchronize(); t0 = time.time(); fn(); torch.cuda.synchronize()
    print(f"{name:28s} {1e3*(time.time()-t0):8.1f} ms  ({H*W/1e6/(time.time()-t0):.0f} Mpixel/s), segments {n}")
