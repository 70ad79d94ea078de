#!/bin/bash
# SQ-side counters of the forward kernel, one rocprofv3 --pmc pass per set (run on the GPU box from the repo root).
# usage: scripts/pmc_fwd_sq.sh <tag> [bench flags...]
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  out=$R/gpurun_out/pmc_${tag}_$name
  timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $out -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-backward $EXTRA > $out.log 2>&1 || return 1
  python3 $R/scripts/pmc_summary.py $out k_fwd >> $R/gpurun_out/pmc_${tag}.txt
}
EXTRA="$*"
rm -f $R/gpurun_out/pmc_${tag}.txt
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM &&
pass b SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR &&
pass c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS &&
pass d SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_CYCLES &&
pass e GRBM_GUI_ACTIVE
echo "pmc_fwd_sq $tag exit $?"
