#!/bin/bash
# usage: scripts/pmc_run.sh TAG PRECISION   (GPU box)
# rocprofv3 of `python3 bench.py --precision P`: one --kernel-trace --stats run, then one --pmc pass per
# counter group (never combined with a trace), summarised per dispatch of the render kernel into
# gpurun_out/TAG_pmc.json; the kernel stats land in gpurun_out/TAG_trace/.
set -e
TAG=$1; PREC=${2:-f16x3}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
B="python3 $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --precision $PREC"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o run -- $B > $OUT/${TAG}_trace.log 2>&1
pass() { n=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/${TAG}_$n -o run -- $B > $OUT/${TAG}_$n.log 2>&1; }
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
pass b SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT
pass c SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_VALU_MFMA_COEXEC_CYCLES
pass d SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU_CVT
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python3 scripts/pmc_summary.py $OUT/${TAG}_a $OUT/${TAG}_b $OUT/${TAG}_c $OUT/${TAG}_d $OUT/${TAG}_fetch $OUT/${TAG}_write > $OUT/${TAG}_pmc.json
cat $OUT/${TAG}_pmc.json
grep -h "nerf_\|Name" $OUT/${TAG}_trace/*kernel_stats.csv | head -8
