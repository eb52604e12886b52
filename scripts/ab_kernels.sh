#!/bin/bash
# usage: bash scripts/ab_kernels.sh a.so b.so ...   (GPU box; names under nerf_amd/csrc/)
# Kernel-level A/B of library builds on the split-precision training step of both networks: one
# `rocprofv3 --kernel-trace --stats` run per build and network, the three MFMA kernels' average durations.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/abk
for lib in "$@"; do
  for net in main legacy; do
    if [ $net = main ]; then B="python3 $R/scripts/bench_train.py 4096 f16x3"; else B="python3 $R/scripts/bench_train_legacy.py 4096 64 f16x3"; fi
    (cd /tmp && NERF_HIP_LIB=$R/nerf_amd/csrc/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abk/${lib}_$net -o run -- $B > $R/gpurun_out/abk/${lib}_$net.log 2>&1)
    echo "== $lib $net"; grep -h "fwd_kernel<true\|fwd_h_kernel<true\|bwd_data\|wgrad\|grad_reduce" $R/gpurun_out/abk/${lib}_$net/*kernel_stats.csv | cut -d, -f1,4 | sed 's/(anonymous namespace):://g' | cut -c1-90
  done
done
