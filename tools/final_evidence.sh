#!/bin/bash
# Round-end evidence beside the artefacts of tools/final_artifacts.sh (one GPU box):
#   PMC passes of the sweep kernels at HEAD, per-wave timelines at both compactness values (needs libobia_hip_tl.so / _tlL.so:
#   tools/build_variant.sh tl -DOBIA_STAMP -DOBIA_ONLY_CP8; tools/build_variant.sh tlL -DOBIA_STAMP -DOBIA_STAMP_KIND=1 -DOBIA_ONLY_CP8),
#   the seam import's cost, and the N = 2 bench line rehearsed over gloo on the one card.
cd "$(dirname "$0")/.."
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tools/r4_pmc_ab.sh - > gpurun_out/fe_pmc_sweeps.txt 2>&1
: > gpurun_out/fe_timelines.txt
for c in 10 0.25; do
  echo "== colour kernels, compactness $c, 10 sweeps per pass (tools/timeline_run.py)" >> gpurun_out/fe_timelines.txt
  OBIA_HIP_LIB=obia_amd/csrc/libobia_hip_tl.so TL_MASK=1 TL_ITERS=10 TL_COMPACT=$c timeout -k 10 200 python tools/timeline_run.py >> gpurun_out/fe_timelines.txt 2>&1
done
echo "== lean pre-pass kernel" >> gpurun_out/fe_timelines.txt
OBIA_HIP_LIB=obia_amd/csrc/libobia_hip_tlL.so TL_MASK=1 TL_ITERS=10 timeout -k 10 200 python tools/timeline_run.py >> gpurun_out/fe_timelines.txt 2>&1
python tools/shard_overhead.py idsk > gpurun_out/fe_shard_overhead.txt 2>&1
python tools/shard_overhead.py ids >> gpurun_out/fe_shard_overhead.txt 2>&1
OBIA_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 2 --size 8192 --steps 3 --warmup 1 --no-cpu --no-side > gpurun_out/fe_bench_gloo2.json 2> gpurun_out/fe_bench_gloo2.err
python tools/golden_exactness.py > gpurun_out/fe_golden_exactness.txt 2>&1
echo evidence done
