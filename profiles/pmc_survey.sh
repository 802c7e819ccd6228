#!/bin/bash
# Stall / busy counters of the step's kernels, one rocprofv3 --pmc pass per counter group (never combined with trace
# domains), towers serialised so that a launch owns the chip.  Run from the repo root through gpurun:
#   bash profiles/pmc_survey.sh r3/pmc   ->  gpurun_out/r3/pmc/<pass>/..., then python3 profiles/pmc_survey.py gpurun_out/r3/pmc
root=$(pwd)
out=$root/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
B="bench.py --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs --steps 2 --warmup 1 --serial-towers"
pass() {
  name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 $B > $out/$name.log 2>&1 || echo "pass $name failed (see $name.log)"
  rm -f $out/$name/*/*_agent_info.csv
  echo "pass $name done"
}
pass sq_wave   SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pass sq_inst   SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS
pass sq_lds    SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS
pass sq_vmem   SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES
pass ta_busy   TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum GRBM_GUI_ACTIVE
pass ta_stall  TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
pass tcp       TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
pass td        TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE
pass spi       SPI_RA_LDS_CU_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_REQ_NO_ALLOC_CSN SPI_RA_RES_STALL_CSN
python3 profiles/pmc_survey.py $out > $out/pmc_survey.txt 2>&1
cat $out/pmc_survey.txt
