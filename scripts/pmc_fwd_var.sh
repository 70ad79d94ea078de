#!/bin/bash
# SQ-side counters of the forward kernel of an experimental build (scripts/exp), one rocprofv3 --pmc pass per set.
# usage: scripts/pmc_fwd_var.sh <variant> "<passes: a b c d e>" [bench flags...]      (on the GPU box, from the repo root)
var=$1; passes=$2; shift; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
EXTRA="$*"
pass() {
  name=$1; shift
  out=$R/gpurun_out/pmc_${var}_$name
  timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $out -- python3 $R/scripts/exp/bench_variant.py $var --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-backward $EXTRA > $out.log 2>&1 || return 1
  python3 $R/scripts/pmc_summary.py $out ${KERN:-k_fwd} >> $R/gpurun_out/pmc_${var}.txt
}
rm -f $R/gpurun_out/pmc_${var}.txt
for p in $passes; do
  case $p in
    a) pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM || exit 1;;
    b) pass b SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR || exit 1;;
    c) pass c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS || exit 1;;
    d) pass d SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_CYCLES || exit 1;;
    e) pass e GRBM_GUI_ACTIVE || exit 1;;
    f) pass f SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_WAVES SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_EXP_GDS || pass f SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES || exit 1;;
    g) pass g SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE || pass g SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES || exit 1;;
    h) pass h FETCH_SIZE || exit 1;;
    i) pass i WRITE_SIZE || exit 1;;
    t1) pass t1 TCP_GATE_EN1 TCP_GATE_EN2 TCP_PENDING_STALL_CYCLES TCP_TOTAL_ACCESSES || exit 1;;
    t2) pass t2 TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_UTCL1_REQUEST TCP_TA_TCP_STATE_READ || exit 1;;
    t3) pass t3 TA_TA_BUSY TA_BUFFER_TOTAL_CYCLES || exit 1;;
    t4) pass t4 TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES || exit 1;;
    t7) pass t7 TA_BUFFER_COALESCED_WRITE_CYCLES TA_BUFFER_COALESCED_READ_CYCLES || exit 1;;
    t5) pass t5 TCP_TCP_TA_DATA_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_WRITE_TAGCONFLICT_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY || exit 1;;
    t6) pass t6 TCP_TAGRAM0_REQ TCP_TAGRAM1_REQ TCP_TCC_WRITE_REQ_LATENCY TCP_TCR_TCP_STALL_CYCLES || exit 1;;
  esac
done
cat $R/gpurun_out/pmc_${var}.txt
