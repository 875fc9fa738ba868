#!/bin/bash
# GPU box: sustained matrix-instruction rates per operand type (tools/mfma_dtype_probe.hip) -> gpurun_out/$1/mfma_dtype_probe.txt
OUT=gpurun_out/${1:-mfma_dtype}; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_dtype_probe tools/mfma_dtype_probe.hip || exit 1
(timeout 120 /tmp/mfma_dtype_probe 60000 0; timeout 120 /tmp/mfma_dtype_probe 60000 1) 2>&1 | tee $OUT/mfma_dtype_probe.txt
