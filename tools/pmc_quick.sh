#!/bin/bash
# usage: tools/pmc_quick.sh <outdir> <workload> -- PMC passes (instruction mix, wait states, LDS)
out=$1; wl=${2:-c2}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
i=0
while read -r set; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python tools/bench_stages.py $wl 3 > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/p$i.log; }
done <<SETS
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_FMA_F64
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
GRBM_GUI_ACTIVE
SETS
python tools/pmc_summary.py $out
