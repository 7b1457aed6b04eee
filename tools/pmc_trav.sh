cd /tmp && export TMPDIR=/tmp; R=/root/repo; mkdir -p $R/gpurun_out/pmc2
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  for v in 0 2; do
    timeout 120 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc2/s${i}_v$v -o out --output-format csv -- python $R/tools/trav_once.py $v 16 1 3 > $R/gpurun_out/pmc2/s${i}_v$v.log 2>&1
  done
done
ls $R/gpurun_out/pmc2
