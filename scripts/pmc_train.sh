#!/bin/bash
# usage: scripts/pmc_train.sh TAG [fp32|f16x3] [main|legacy] [hidden_size encoding_size]   (GPU box)
# rocprofv3 of one training step (scripts/bench_train.py, or bench_train_legacy.py for the 8 x 256 network of
# examples/nerf.pth; 4096 rays x 64, arithmetic as given):
# one --kernel-trace --stats run, then one --pmc pass per counter group (never combined with a trace),
# summarised per dispatch for the three MFMA kernels of the step into gpurun_out/TAG_{fwd,dgrad,wgrad}_pmc.json.
set -e
TAG=$1; PREC=${2:-fp32}; NET=${3:-main}; HID=${4:-256}; ENC=${5:-32}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
B="python3 $PWD/scripts/bench_train.py 4096 $PREC $HID $ENC"
FWD="nerf_render_fwd_kernel<true"; DG=nerf_bwd_data_; WG=nerf_wgrad_
if [ "$NET" = legacy ]; then
  B="python3 $PWD/scripts/bench_train_legacy.py 4096 64 $PREC"
  FWD="nerf_legacy_fwd_"; DG=nerf_legacy_bwd_data_; WG=nerf_legacy_wgrad_
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o run -- $B > $OUT/${TAG}_trace.log 2>&1
pass() { n=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/${TAG}_$n -o run -- $B > $OUT/${TAG}_$n.log 2>&1; }
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
pass d SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS
pass b SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_CVT SQ_IFETCH SQ_ACTIVE_INST_SCA
pass fetch FETCH_SIZE
pass write WRITE_SIZE
summ() { python3 scripts/pmc_summary.py $OUT/${TAG}_a $OUT/${TAG}_d $OUT/${TAG}_b $OUT/${TAG}_fetch $OUT/${TAG}_write --kernel "$2" > $OUT/${TAG}_$1_pmc.json; }
summ fwd "$FWD"
summ dgrad $DG
summ wgrad $WG
grep -h "nerf_\|Name" $OUT/${TAG}_trace/*kernel_stats.csv | head -12
