#!/bin/bash
# Developer tool (GPU box): fresh step timeline + PMC counters of the kernels around the sweeps (features, connectivity, tiler) at 16384^2.
cd "$(dirname "$0")/.."
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ot_step -- python3 tools/step_trace.py > gpurun_out/ot_s.log 2>&1
python3 tools/trace_gaps.py gpurun_out/ot_step > gpurun_out/ot_step_timeline.txt
rm -f gpurun_out/ot_step/*/*kernel_trace.csv
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $set --output-format csv -d gpurun_out/opmc/$i -- python3 tools/step_trace.py > gpurun_out/opmc_$i.log 2>&1 || echo "set $i failed"
done
for k in features_planes band_minmax cc_tile cc_flatten cc_relabel tile_scatter cc_seam ids_apply tile_mask zonal_kernel; do python3 tools/pmc_summary.py gpurun_out/opmc $k; done > gpurun_out/opmc_summary.txt
cat gpurun_out/ot_step_timeline.txt
