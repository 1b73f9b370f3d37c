import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from obia_amd.tiling import create_tiled_segments
from bench import synth_raster
dev = torch.device('cuda:0')
S = int(os.environ.get('DBG_SIZE', '4096')); T = int(os.environ.get('DBG_TILE', '2048'))
img = synth_raster(S, S, 8, seed=0, device=dev)
mask = torch.ones((S, S), dtype=torch.uint8, device=dev)
lab, n = create_tiled_segments(img, input_mask=mask, tile_size=T, buffer=64, crown_radius=5, pixel_size=(0.5, 0.5), compactness=10.0)
print(os.environ.get('OBIA_PREP_GROUPED'), n, int(lab.sum().item()), int((lab.long() * torch.arange(lab.numel(), device=dev).view_as(lab) % 1000003).sum().item()))
