#!/bin/bash
# SQ / LDS / TCP counters of the backward brick kernel (shipped library or an experimental build), one rocprofv3 --pmc pass per set.
# usage: scripts/pmc_bwd_var.sh <variant|shipped> "<passes: a b c t1 t2 e>"      (on the GPU box, from the repo root)
var=$1; passes=$2; shift; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  out=$R/gpurun_out/pmcb_${var}_$name
  timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $out -- python3 $R/scripts/exp/bench_variant.py $var --steps 1 --warmup 1 --no-cpu-baseline --no-check > $out.log 2>&1 || return 1
  python3 $R/scripts/pmc_summary.py $out k_bwd_brick >> $R/gpurun_out/pmcb_${var}.txt
}
rm -f $R/gpurun_out/pmcb_${var}.txt
for p in $passes; do
  case $p in
    a) pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM || exit 1;;
    b) pass b SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR || exit 1;;
    c) pass c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS || exit 1;;
    e) pass e GRBM_GUI_ACTIVE || exit 1;;
    h) pass h FETCH_SIZE || exit 1;;
    i) pass i WRITE_SIZE || exit 1;;
    t1) pass t1 TCP_GATE_EN1 TCP_GATE_EN2 TCP_PENDING_STALL_CYCLES TCP_TOTAL_ACCESSES || exit 1;;
    t2) pass t2 TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_TCC_ATOMIC_WITHOUT_RET_REQ TCP_TA_TCP_STATE_READ || exit 1;;
    d) pass d SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_CYCLES || exit 1;;
  esac
done
cat $R/gpurun_out/pmcb_${var}.txt
