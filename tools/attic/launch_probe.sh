#!/bin/bash
# via gpurun: kernel-boundary / grid-barrier price list of this box + the single-query search under runtime knobs
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-launch_probe}; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 120 tools/attic/launch_probe > $O/probe_default.txt 2>&1
HIP_FORCE_DEV_KERNARG=1 timeout 120 tools/attic/launch_probe > $O/probe_devkernarg.txt 2>&1
HIP_FORCE_DEV_KERNARG=0 timeout 120 tools/attic/launch_probe > $O/probe_hostkernarg.txt 2>&1
B="python bench.py --batch 1 --steps 20 --warmup 3 --no-cpu-baseline --no-exact-fp32 --no-roofline --secondary ''"
run() { local tag=$1; shift; env "$@" bash -c "$B" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],3), 'ms')"; }
{
run default A=1
run dev_kernarg HIP_FORCE_DEV_KERNARG=1
run host_kernarg HIP_FORCE_DEV_KERNARG=0
run graph_pkt_capture_off DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run graph_pkt_capture_on DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run graph_batch_64 DEBUG_HIP_GRAPH_BATCH_SIZE=64
run sys_scope_signal_0 ROC_SYSTEM_SCOPE_SIGNAL=0
B="$B --no-graph"
run eager A=1
run eager_dev_kernarg HIP_FORCE_DEV_KERNARG=1
} > $O/bench_q1_knobs.txt 2>&1
cat $O/probe_default.txt $O/bench_q1_knobs.txt
