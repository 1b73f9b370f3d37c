"""Exercise the sharded driver's RCCL code path with a one-rank nccl group on a single GPU (all_gather / barrier /
all_reduce on device tensors).  The multi-rank exchange itself needs several GPUs; this only proves the backend wiring."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
import torch, torch.distributed as dist
import bench
from obia_amd import _lib
from obia_amd.distributed import ShardedTiler
from obia_amd.tiling import create_tiled_segments
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
H, W, C = 1024, 2048, 8
img = bench.synth_raster(H, W, C, seed=0, device=dev, row0=0)
mask = torch.ones((H, W), dtype=torch.uint8, device=dev)
ctx = _lib.Context(0)
t = ShardedTiler(img, mask, H, H // 512, 512, 32, 5, (0.5, 0.5), ctx=ctx, compactness=10.0)
lab, n = t.run()
t.close()
ref, n_ref = create_tiled_segments(img, input_mask=mask, tile_size=512, buffer=32, crown_radius=5, pixel_size=(0.5, 0.5),
                                   compactness=10.0, white_order="parity", ctx=ctx)
x = torch.tensor([1.0], device=dev); dist.all_reduce(x); dist.barrier()
print("nccl one-rank ok:", n, n_ref, bool(torch.equal(lab, ref)))
dist.destroy_process_group()
