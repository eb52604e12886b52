#!/bin/bash
# usage: scripts/pmc_narrow.sh TAG HIDDEN ENC PRECISION   (GPU box)
# rocprofv3 of `python3 scripts/bench_narrow.py HIDDEN ENC PRECISION` (the 800x800x128 frame through the kernel
# instantiated for that width): one --kernel-trace --stats run, then one --pmc pass per counter group (never combined
# with a trace), summarised per dispatch of the render kernel into gpurun_out/TAG_pmc.json.
set -e
TAG=$1; HID=${2:-128}; ENC=${3:-32}; PREC=${4:-fp32}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
B="python3 $PWD/scripts/bench_narrow.py $HID $ENC $PREC"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o run -- $B > $OUT/${TAG}_trace.log 2>&1
pass() { n=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/${TAG}_$n -o run -- $B > $OUT/${TAG}_$n.log 2>&1; }
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
pass d SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python3 scripts/pmc_summary.py $OUT/${TAG}_a $OUT/${TAG}_d $OUT/${TAG}_fetch $OUT/${TAG}_write > $OUT/${TAG}_pmc.json
cat $OUT/${TAG}_pmc.json
grep -h "nerf_\|Name" $OUT/${TAG}_trace/*kernel_stats.csv | head -4
