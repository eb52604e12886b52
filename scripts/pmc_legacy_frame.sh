#!/bin/bash
# usage: scripts/pmc_legacy_frame.sh TAG   (GPU box)
# rocprofv3 of `python3 scripts/bench_legacy_frame.py 1`: one --kernel-trace --stats run, then SQ / FETCH / WRITE passes (never
# combined with a trace), summarised per dispatch of nerf_legacy_fwd_kernel<false> into gpurun_out/TAG_pmc.json.
set -e
TAG=$1
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
B="python3 $PWD/scripts/bench_legacy_frame.py 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o run -- $B > $OUT/${TAG}_trace.log 2>&1
pass() { n=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/${TAG}_$n -o run -- $B > $OUT/${TAG}_$n.log 2>&1; }
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
pass d SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python3 scripts/pmc_summary.py $OUT/${TAG}_a $OUT/${TAG}_d $OUT/${TAG}_fetch $OUT/${TAG}_write --kernel "nerf_legacy_fwd_kernel<false" > $OUT/${TAG}_pmc.json
cat $OUT/${TAG}_pmc.json
grep -h "nerf_\|Name" $OUT/${TAG}_trace/*kernel_stats.csv | head -4
