"""Developer probe: the zonal-statistics pass alone on the headline raster (16384 x 16384 x C) with the label raster the tiler makes
of it.  Prints the HIP-event time of `zonal_kernel` per call.  OBIA_HIP_LIB selects a variant library (tools/build_variant.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from obia_amd import _lib
from obia_amd.statistics import zonal_stats
from obia_amd.tiling import create_tiled_segments
from bench import synth_raster

S = int(os.environ.get("ZT_SIZE", "16384"))
C = int(os.environ.get("ZT_BANDS", "8"))
dev = torch.device("cuda:0")
img = synth_raster(S, S, C, seed=0, device=dev)
mask = torch.ones((S, S), dtype=torch.uint8, device=dev)
ctx = _lib.default_context(0)
ctx.set_profiling(1)
lab, n = create_tiled_segments(img, input_mask=mask, tile_size=2048, buffer=64, crown_radius=5, pixel_size=(0.5, 0.5), compactness=10.0, ctx=ctx)
ts = []
for rep in range(8):
    st = zonal_stats(img, lab, n_labels=n, ctx=ctx)
    torch.cuda.synchronize()
    ts.append(ctx.timing()["zonal_ms"])
ts = sorted(ts[2:])
chk = float(torch.nan_to_num(st["mean"]).sum().item()) if isinstance(st, dict) and "mean" in st else 0.0
print(f"{os.environ.get('OBIA_HIP_LIB', 'default')}: zonal {ts[len(ts)//2]:.3f} ms (min {ts[0]:.3f}) segments {n} -> {S*S*(4*C+4)/ts[len(ts)//2]/1e6:.0f} GB/s  checksum {chk:.6e}", flush=True)
