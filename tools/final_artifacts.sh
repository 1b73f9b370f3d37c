#!/bin/bash
# Round-end artefacts on one GPU box: bench lines, kernel statistics of the same command, PMC passes for the traffic file.
#   gpurun -- 'bash tools/final_artifacts.sh'   then   python tools/traffic_json.py gpurun_out/fa_fetch gpurun_out/fa_write <commit>
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python bench.py > gpurun_out/fa_bench.json 2> gpurun_out/fa_bench.err
# (the profiled command is the timed workload alone: with the side legs the statistics would average the colour sweep over the
# 16384^2 and the 32768^2 rasters, whose launches differ in size)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fa_trace -- python3 bench.py --no-cpu --no-side > gpurun_out/fa_bench_rocprof.json 2> gpurun_out/fa_rocprof.err
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/fa_fetch -- python3 tools/step_trace.py > gpurun_out/fa_f.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/fa_write -- python3 tools/step_trace.py > gpurun_out/fa_w.log 2>&1
timeout -k 10 400 python bench.py --config c4 --no-cpu --no-side > gpurun_out/fa_bench_c4.json 2> gpurun_out/fa_c4.err
timeout -k 10 200 python bench.py --bands 3 --no-cpu --no-side --steps 3 --warmup 1 > gpurun_out/fa_bench_bands3.json 2> gpurun_out/fa_b3.err
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fa_step -- python3 tools/step_trace.py > gpurun_out/fa_s.log 2>&1
python3 tools/trace_gaps.py gpurun_out/fa_step > gpurun_out/fa_step_timeline.txt
OBIA_TRACE_COMPACTNESS=0.25 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fa_step025 -- python3 tools/step_trace.py > gpurun_out/fa_s025.log 2>&1
python3 tools/trace_gaps.py gpurun_out/fa_step025 > gpurun_out/fa_step_timeline_c025.txt
rm -f gpurun_out/fa_trace/*/*kernel_trace.csv gpurun_out/fa_step/*/*kernel_trace.csv gpurun_out/fa_step025/*/*kernel_trace.csv   # (tens of MB; the statistics are what is kept)
echo artefacts done
