#!/bin/bash
# Developer tool (GPU box): PMC counters of the sweep kernels over two steps of the headline workload at 8192^2 (counter passes only,
# no tracing domains); summary to gpurun_out/spmc_summary.txt
cd "$(dirname "$0")/.."
export OBIA_TRACE_SIZE=${OBIA_TRACE_SIZE:-8192}
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INSTS_FLAT" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_IFETCH" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $set --output-format csv -d gpurun_out/spmc/$i -- python3 tools/step_trace.py > gpurun_out/spmc_$i.log 2>&1 || echo "set $i failed"
done
python3 tools/pmc_summary.py gpurun_out/spmc "${1:-slic_}" > gpurun_out/spmc_summary.txt
cat gpurun_out/spmc_summary.txt
