#!/bin/bash
# A/B of several builds of the library inside ONE gpurun call (same box): two interleaved rounds of `bench.py --no-cpu --no-side`.
#   tools/ab.sh <suffix> <suffix> ... [-- bench args]      ("-" = obia_amd/csrc/libobia_hip.so, "old" = libobia_hip_old.so, ...)
cd "$(dirname "$0")/.."
libs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done
[ "$1" = "--" ] && shift
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; s=d["stage_ms_per_step"]; print("step %.2f ms  sweep %.4f ms/launch  frac %.3f  prepass %.2f  assign %.2f  cc %.2f  feat %.2f  zonal %.2f  segs %d" % (d["ms_per_step"], r["avg_launch_ms"], r["frac"], s["prepass_ms"], s["assign_ms"], s["connectivity_ms"], s["features_ms"], s["zonal_ms"], d["config"]["segments"]))'
for rep in 1 2; do
  for v in "${libs[@]}"; do
    if [ "$v" = "-" ]; then lib=obia_amd/csrc/libobia_hip.so; else lib=obia_amd/csrc/libobia_hip_$v.so; fi
    printf "%-8s %s : " "${v}" "$*"
    OBIA_HIP_LIB=$lib timeout -k 10 200 python bench.py --no-cpu --no-side --steps 6 "$@" 2>gpurun_out/ab_err.txt | python -c "$pick" || tail -3 gpurun_out/ab_err.txt
  done
done
