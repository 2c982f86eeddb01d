#!/bin/bash
# Kernel timeline of the last frames of a bench run: rocprofv3 --kernel-trace, every launch with start and duration in launch order
# (tools/trace_depths.py).   usage (through gpurun): bash tools/timeline.sh [workload] [launches] [out.txt]
export TMPDIR=/tmp
WL=${1:-hf10m}; N=${2:-160}; OUT=${3:-gpurun_out/timeline_$WL.txt}
D=gpurun_out/kt_tl
rm -rf $D
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/tl.log 2>&1 || { tail -5 gpurun_out/tl.log; exit 1; }
python3 tools/trace_depths.py "$(find $D -name '*kernel_trace.csv' | head -1)" $N > $OUT
rm -rf $D
