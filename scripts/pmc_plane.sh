#!/bin/bash
# SQ / TA counters of the plane backward at BASELINE configs[1] (GPU box, repo root)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -f $R/gpurun_out/pmc_plane.txt
pass() {
  name=$1; shift
  out=$R/gpurun_out/pmc_plane_$name
  timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $out -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check --batch 8 --grid 32 --channels 256 --views 4 > $out.log 2>&1 || return 1
  python3 $R/scripts/pmc_summary.py $out k_plane_ds >> $R/gpurun_out/pmc_plane.txt; python3 $R/scripts/pmc_summary.py $out k_bwd_plane >> $R/gpurun_out/pmc_plane.txt
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM &&
pass b SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL &&
pass c TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum &&
pass d TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
cat $R/gpurun_out/pmc_plane.txt
