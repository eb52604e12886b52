#!/bin/bash
# usage: scripts/pmc_run.sh TAG PRECISION   (GPU box; one rocprofv3 --pmc pass per counter group)
set -e
TAG=$1; PREC=${2:-fp32}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
B="python3 $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --precision $PREC"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_a -o run -- $B > $OUT/${TAG}_a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/${TAG}_b -o run -- $B > $OUT/${TAG}_b.log 2>&1
rocprofv3 --pmc SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $OUT/${TAG}_c -o run -- $B > $OUT/${TAG}_c.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU_CVT --output-format csv -d $OUT/${TAG}_d -o run -- $B > $OUT/${TAG}_d.log 2>&1
python3 scripts/pmc_summary.py $OUT/${TAG}_a $OUT/${TAG}_b $OUT/${TAG}_c $OUT/${TAG}_d > $OUT/${TAG}_pmc.json
cat $OUT/${TAG}_pmc.json
