#!/bin/bash
# via gpurun: decoder-layer GEMM chain at 400 .. 1300 rows, default routes vs the wave-split tiles beyond 768 rows
export RPR_DEV_LIB=1   # the switches below are development switches: libripor_hip_dev.so (same sources, -DRPR_DEV_SWITCHES)
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r06_gemm_mid}; mkdir -p $O
cd $GRAFT_REPO_ROOT
MS="400 640 1000 1300"
run() { local tag=$1; shift; env "$@" python tools/gemm_bench.py $MS > $O/$tag.txt 2>&1; grep -h "M=" $O/$tag.txt | sed "s/^/$tag: /"; }
run default A=1
run wsplit1400 RPR_GEMM_WSPLIT_MAX=1400
run wsplit1400_cfg2 RPR_GEMM_WSPLIT_MAX=1400 RPR_WSPLIT_CFG=2 RPR_WSPLIT_KS=1
run wsplit1400_cfg2_ks2 RPR_GEMM_WSPLIT_MAX=1400 RPR_WSPLIT_CFG=2 RPR_WSPLIT_KS=2
