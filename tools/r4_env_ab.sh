#!/bin/bash
# A/B of an environment switch on the same library inside one gpurun call:  tools/r4_env_ab.sh VAR "0 1" [bench args]
cd "$(dirname "$0")/.."
var=$1; vals=$2; shift 2
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; s=d["stage_ms_per_step"]; print("step %.2f ms  sweep %.4f ms/launch  frac %.3f  prepass %.2f  assign %.2f  cc %.2f  feat %.2f  zonal %.2f  segs %d" % (d["ms_per_step"], r["avg_launch_ms"], r["frac"], s["prepass_ms"], s["assign_ms"], s["connectivity_ms"], s["features_ms"], s["zonal_ms"], d["config"]["segments"]))'
for rep in 1 2; do
  for v in $vals; do
    printf "%s=%-4s %s : " "$var" "$v" "$*"
    env $var=$v timeout -k 10 200 python bench.py --no-cpu --no-side --steps 6 "$@" 2>gpurun_out/ab_err.txt | python -c "$pick" || tail -3 gpurun_out/ab_err.txt
  done
done
