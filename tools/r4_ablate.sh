#!/bin/bash
# Round 4, first call: the ablation builds of the sweep body at both compactness values, and the per-wave timelines.
#   tools/build_variant.sh noload -DOBIA_ABL_NOLOAD; ... v1 -DOBIA_ABL_VISITS=1; ... noacc -DOBIA_ABL_NOACC; ... tl -DOBIA_STAMP; ... tlL -DOBIA_STAMP -DOBIA_STAMP_KIND=1
#   gpurun -- 'bash tools/r4_ablate.sh'
cd "$(dirname "$0")/.."
out=gpurun_out/r4_ablate.txt
: > $out
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; s=d["stage_ms_per_step"]; print("step %.2f ms  sweep %.4f ms/launch  frac %.3f  prepass %.2f  assign %.2f  cc %.2f" % (d["ms_per_step"], r["avg_launch_ms"], r["frac"], s["prepass_ms"], s["assign_ms"], s["connectivity_ms"]))'
for comp in 10 0.25; do
  for v in "" _noload _v1 _noacc; do
    lib=obia_amd/csrc/libobia_hip$v.so
    echo "== compactness $comp lib ${v:-HEAD}" >> $out
    OBIA_HIP_LIB=$lib timeout -k 10 200 python bench.py --no-cpu --no-side --steps 6 --compactness $comp 2>gpurun_out/r4_abl_err.txt | python -c "$pick" >> $out 2>&1
  done
done
for comp in 10 0.25; do
  echo "== timeline colour kernels, compactness $comp" >> $out
  OBIA_HIP_LIB=obia_amd/csrc/libobia_hip_tl.so TL_MASK=1 TL_ITERS=4 TL_COMPACT=$comp timeout -k 10 200 python tools/timeline_run.py >> $out 2>&1
done
echo "== timeline lean pre-pass kernel" >> $out
OBIA_HIP_LIB=obia_amd/csrc/libobia_hip_tlL.so TL_MASK=1 TL_ITERS=4 timeout -k 10 200 python tools/timeline_run.py >> $out 2>&1
cat $out
