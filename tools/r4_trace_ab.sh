#!/bin/bash
# One warm step under the kernel trace for several builds: where the step's time goes, kernel by kernel.   tools/r4_trace_ab.sh - old
cd "$(dirname "$0")/.."
for v in "$@"; do
  if [ "$v" = "-" ]; then lib=obia_amd/csrc/libobia_hip.so; else lib=obia_amd/csrc/libobia_hip_$v.so; fi
  rm -rf gpurun_out/trace_$v
  OBIA_HIP_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_$v -- python3 tools/step_trace.py > gpurun_out/trace_$v.log 2>&1
  echo "=== build $v"; python3 tools/trace_gaps.py gpurun_out/trace_$v
  rm -f gpurun_out/trace_$v/*/*kernel_trace.csv
done
