#!/bin/bash
# usage: scripts/pmc_train.sh TAG KERNEL_SUBSTR   (GPU box) — SQ counters of one kernel of the training step
set -e
TAG=$1; KERN=${2:-nerf_wgrad_kernel}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
B="python3 $PWD/scripts/bench_train.py 4096"
pass() { n=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/${TAG}_$n -o run -- $B > $OUT/${TAG}_$n.log 2>&1; }
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
pass d SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS
python3 scripts/pmc_summary.py $OUT/${TAG}_a $OUT/${TAG}_d --kernel $KERN > $OUT/${TAG}_pmc.json
cat $OUT/${TAG}_pmc.json
