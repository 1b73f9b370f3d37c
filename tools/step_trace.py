"""One warm step of the headline workload (tiled SLIC + zonal statistics) for rocprofv3 --kernel-trace:
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 tools/step_trace.py
then  python3 tools/trace_gaps.py gpurun_out/trace  prints where the time between kernels goes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_raster
from obia_amd import _lib
from obia_amd.statistics import zonal_stats
from obia_amd.tiling import create_tiled_segments
S = int(os.environ.get("OBIA_TRACE_SIZE", "16384"))
dev = torch.device("cuda", 0)
img = synth_raster(S, S, int(os.environ.get("OBIA_TRACE_BANDS", "8")), 0, dev)
mask = torch.ones((S, S), dtype=torch.uint8, device=dev)
ctx = _lib.Context(0)
kw = dict(tile_size=2048, buffer=64, crown_radius=5, pixel_size=(0.5, 0.5), compactness=float(os.environ.get("OBIA_TRACE_COMPACTNESS", "10")), ctx=ctx)
for _ in range(int(os.environ.get("OBIA_TRACE_STEPS", "2"))):
    lab, n = create_tiled_segments(img, input_mask=mask, **kw)
    st = zonal_stats(img, lab, n_labels=n, ctx=ctx)
torch.cuda.synchronize()
print("segments", n)
