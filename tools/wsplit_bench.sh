#!/bin/bash
# via gpurun: per-shape launch times of the projection GEMMs at 80 .. 1500 rows, old routes vs wave-split tile shapes
export RPR_DEV_LIB=1   # the switches below are development switches: libripor_hip_dev.so (same sources, -DRPR_DEV_SWITCHES)
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-wsplit}; mkdir -p $O
cd $GRAFT_REPO_ROOT
MS="80 160 280 640 1000 1500"
run() { local tag=$1; shift; env "$@" python tools/gemm_bench.py $MS > $O/$tag.txt 2>&1; grep -h "weighted" $O/$tag.txt | sed "s/^/$tag: /"; }
run old RPR_GEMM_WSPLIT_MAX=0
run auto A=1
run cfg0 RPR_WSPLIT_CFG=0 RPR_WSPLIT_KS=1
run cfg1 RPR_WSPLIT_CFG=1 RPR_WSPLIT_KS=1
run cfg2 RPR_WSPLIT_CFG=2 RPR_WSPLIT_KS=1
run cfg1_ks2 RPR_WSPLIT_CFG=1 RPR_WSPLIT_KS=2
run cfg2_ks2 RPR_WSPLIT_CFG=2 RPR_WSPLIT_KS=2
run cfg2_ks4 RPR_WSPLIT_CFG=2 RPR_WSPLIT_KS=4
