set -e
cd "$(dirname "$0")/.."
export ZT_SIZE=8192
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" "SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_IFETCH SQ_INST_LEVEL_LDS" "FETCH_SIZE" "SQ_INST_LEVEL_VMEM SQ_LDS_ATOMIC_RETURN SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $set --output-format csv -d gpurun_out/zpmc/$i -- python3 tools/zonal_time.py > gpurun_out/zpmc_$i.log 2>&1 || echo "set $i failed"
done
python3 tools/pmc_summary.py gpurun_out/zpmc zonal_kernel > gpurun_out/zpmc_summary.txt
cat gpurun_out/zpmc_summary.txt
