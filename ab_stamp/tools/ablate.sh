#!/bin/bash
# timing experiments: OBIA_ABLATE bits (1 = no accumulation, 2 = no LDS->global flush, 4 = no candidate evaluation)
for a in 0 1 2 4 5; do
  echo "== OBIA_ABLATE=$a"
  OBIA_ABLATE=$a python tools/gpu_time.py 2>&1 | grep -v amdgpu.ids | head -3
done
