#!/bin/bash
# PMC counters of the sweep kernels for two builds of the library (counter passes only): tools/r4_pmc_ab.sh <suffix-A> <suffix-B>
cd "$(dirname "$0")/.."
export OBIA_TRACE_SIZE=${OBIA_TRACE_SIZE:-8192}
for v in "$@"; do
  if [ "$v" = "-" ]; then lib=obia_amd/csrc/libobia_hip.so; else lib=obia_amd/csrc/libobia_hip_$v.so; fi
  export OBIA_HIP_LIB=$lib
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_SMEM"; do
    i=$((i+1))
    timeout -k 10 150 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmcab_$v/$i -- python3 tools/step_trace.py > gpurun_out/pmcab_${v}_$i.log 2>&1 || echo "set $i failed"
  done
  echo "=== build $v" ; python3 tools/pmc_summary.py gpurun_out/pmcab_$v "${PMC_FILTER:-slic_}"
done
