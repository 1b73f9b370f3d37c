"""Time the polygon pass on the bench raster's label map (16384^2, ~826 k segments)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from obia_amd import _lib
from obia_amd.tiling import create_tiled_segments
from obia_amd.polygons import polygonize
H = W = int(os.environ.get("SIZE", 16384))
img = bench.synth_raster(H, W, 8, seed=0, device=torch.device("cuda", 0), row0=0)
mask = torch.ones((H, W), dtype=torch.uint8, device="cuda")
lab, n = create_tiled_segments(img, input_mask=mask, tile_size=2048, buffer=64, crown_radius=5, pixel_size=(0.5, 0.5), compactness=10.0)
del img
lib = _lib.load()
import ctypes
c = _lib.default_context(0)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    nr, nv = ctypes.c_int64(0), ctypes.c_int64(0)
    _lib.check(lib.obia_polygon_count_i32_dev(c.handle, lab.data_ptr(), H, W, 1, ctypes.byref(nr), ctypes.byref(nv)))
    t1 = time.time()
    print(f"count pass: {1e3*(t1-t0):.1f} ms  rings {nr.value} vertices {nv.value}")
t0 = time.time()
tab = polygonize(lab, affine_transformation=[0.5, 0, 0, -0.5, 0, 0], start_label=1)
print(f"polygonize (GPU passes + host grouping): {time.time()-t0:.2f} s, polygons {len(tab)}, segments {n}")
