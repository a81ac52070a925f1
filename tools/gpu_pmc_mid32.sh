#!/bin/bash
# SQ counters of the mid-size product kernel (separate passes, counters only with --kernel-trace)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_mid32; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pass$i -- python3 tools/pmc_mid32.py > /dev/null 2> $O/err$i.txt || { tail -3 $O/err$i.txt; }
done
python tools/pmc_mid32_table.py $O > $O/table.md; cat $O/table.md
