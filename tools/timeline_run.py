"""Diagnostic: per-wave timeline of the SLIC colour sweep (the sweep before the last of one call) on one 4096^2 x 8 raster.
   tools/build_variant.sh tl -DOBIA_STAMP                                   # stamps on, records per wave, no atomics
   OBIA_HIP_LIB=obia_amd/csrc/libobia_hip_tl.so python tools/timeline_run.py
   tools/build_variant.sh tl1 -DOBIA_STAMP -DOBIA_ABL_LDSPAD=110000          # + LDS ballast: ONE workgroup per CU (pure chain latency)
Environment: TL_ITERS (sweeps per call, default 4), TL_MASK=1 (masked path: pre-pass + colour pass), TL_COMPACT (compactness, default 10).
Prints phase cycles per wave (s_memtime ticks = core clock, checked against the 100-MHz clock), the wave lifetime and the visit counters.
Other ablation switches of slic_sweep.hip: -DOBIA_ABL_NOLOAD (no feature / mask traffic), -DOBIA_ABL_VISITS=n (at most n visits per
footprint), -DOBIA_ABL_NOACC (no centroid update).  Numbers: profiles/r02_pmc_notes.md."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from obia_amd import _lib
from obia_amd.segmentation import slic
_lib.load()
H = W = 4096; C = 8
g = torch.Generator(device="cuda").manual_seed(0)
yy = torch.arange(H, device="cuda", dtype=torch.float32)[:, None]
xx = torch.arange(W, device="cuda", dtype=torch.float32)[None, :]
img = torch.empty((H, W, C), device="cuda", dtype=torch.float32)
for c in range(C):
    img[:, :, c] = 400.0 * torch.sin(xx / (11 + 3 * c)) * torch.cos(yy / (13 + 2 * c)) + 1000 + 50 * c + 20.0 * torch.randn((H, W), device="cuda", generator=g)
n = round(H * W / 324.0)
iters = int(os.environ.get("TL_ITERS", "4"))   # the last sweep of a call does not accumulate: look at the one before by running iters+... no: max_num_iter
mask = torch.ones((H, W), dtype=torch.bool, device="cuda") if os.environ.get("TL_MASK") else None
for _ in range(2):
    slic(img, n_segments=n, compactness=float(os.environ.get("TL_COMPACT", "10")), _normalize_bands=True, max_num_iter=iters, mask=mask, _stage="pre")
torch.cuda.synchronize()
NT = (H // 64) * (W // 64)
buf = np.zeros(NT * 4 * 24, np.uint64)
ctypes.CDLL(_lib.LIB_PATH).obia_debug_timeline(buf.ctypes.data_as(ctypes.c_void_p), NT * 4)
r = buf.reshape(NT, 4, 24).astype(np.float64)
names = ["sort", "prologue+stage", "scoring", "visits: loop control", "labels", "run merge", "fold(+fetch issue)", "barrier+flush", "visit: record+next min", "visit: spatial+live", "visit: colours+keys", "-"]
life = r[:, :, 13] - r[:, :, 12]
rt = (r[:, :, 15] - r[:, :, 14]).mean() * 10.0
print('wave lifetime %.0f ns on the 100-MHz clock -> %.3f ticks per ns' % (rt, life.mean() / rt))
print("per wave: lifetime %.0f ticks" % life.mean())
for i in [1, 0, 2, 3, 8, 9, 10, 4, 5, 6, 7]:
    print("  %-20s %9.0f  %5.1f %%" % (names[i], r[:, :, i].mean(), 100 * r[:, :, i].mean() / life.mean()))
wg0 = r[:, :, 12].min(axis=1); wg1 = r[:, :, 13].max(axis=1)
span = wg1.max() - wg0.min()
print("workgroup lifetime %.0f ticks; kernel span %.0f ticks; workgroups alive on average %.1f (= %.2f per CU)" %
      ((wg1 - wg0).mean(), span, (wg1 - wg0).sum() / span, (wg1 - wg0).sum() / span / 256))
print("footprints/wave %.2f visits/footprint %.2f colour visits %.2f pair evals %.2f" %
      (r[:, :, 16].mean(), r[:, :, 17].sum() / r[:, :, 16].sum(), r[:, :, 18].sum() / r[:, :, 16].sum(), r[:, :, 19].sum() / r[:, :, 16].sum()))
