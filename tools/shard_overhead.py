"""Host-overhead probe: the sharded driver with ONE rank vs the plain tiled driver on the same slab."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("gloo", rank=0, world_size=1)
from bench import synth_raster
from obia_amd import _lib
from obia_amd.tiling import create_tiled_segments
from obia_amd.distributed import ShardedTiler
H = W = int(os.environ.get("SIZE", 16384)); C = 8
img = synth_raster(H, W, C, 0, torch.device("cuda"))
mask = torch.ones((H, W), dtype=torch.uint8, device="cuda")
ctx = _lib.Context(0)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    lab, n = create_tiled_segments(img, input_mask=mask, tile_size=2048, buffer=64, crown_radius=5, pixel_size=(0.5, 0.5), white_order="parity", ctx=ctx)
    torch.cuda.synchronize(); t1 = time.time()
    t = ShardedTiler(img, mask, H, H // 2048, 2048, 64, 5, (0.5, 0.5), ctx=ctx, compactness=10.0)
    torch.cuda.synchronize(); t2 = time.time()
    lab2, n2 = t.run()
    torch.cuda.synchronize(); t3 = time.time()
    ext, dense, no = t.owned_labels(); t.close()
    torch.cuda.synchronize(); t4 = time.time()
    print(f"plain {1e3*(t1-t0):.1f} ms | sharded: setup {1e3*(t2-t1):.1f} run {1e3*(t3-t2):.1f} owned {1e3*(t4-t3):.1f} | n {n} {n2} {no}", flush=True)
dist.destroy_process_group()
