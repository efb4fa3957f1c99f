#!/bin/bash
# usage: tools/pmc.sh <outdir> <workload> -- runs the PMC passes one at a time (rocprofv3 --pmc only)
out=$1; wl=${2:-c2}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
i=0
while read -r set; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python tools/bench_stages.py $wl 3 > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/p$i.log; }
done <<SETS
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
FETCH_SIZE
WRITE_SIZE
GRBM_GUI_ACTIVE
TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
SQ_INSTS_VALU_FMA_F64 SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_WAVES_LT_64 SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
SETS
python tools/pmc_summary.py $out
