#!/bin/bash
# One profiling round on the GPU box: rocprofv3 PMC passes (separate runs) and kernel-trace stats for the headline workload
# (hf10m = BASELINE configs[3]) and the incoherent one (soup1m = configs[2]), then the un-profiled bench lines; summaries
# land in profiles/ under the given tag (and are mirrored to gpurun_out/profiles_<tag>/ so they come back from the box).
#   usage: tools/profile_round.sh <tag>        (run through gpurun; every GPU step has its own time limit)
set -u
TAG=${1:-r03}
export TMPDIR=/tmp
mkdir -p gpurun_out profiles
for WL in hf10m soup1m; do
  STEPS=3; [ $WL = hf10m ] && STEPS=1
  # under rocprofv3 --pmc kernels run one at a time and slower, which could tip the traversal-mode calibration the other way:
  # the profiled runs are pinned to the mode the un-profiled bench measures for the workload (bench line: config.refill)
  export RTHIP_REFILL_MODE=0; [ $WL = hf10m ] && export RTHIP_REFILL_MODE=2
  bash tools/profile_pmc.sh $TAG $WL $STEPS > gpurun_out/${TAG}_${WL}_pmc.log 2>&1 || { echo "pmc $WL failed"; tail -5 gpurun_out/${TAG}_${WL}_pmc.log; exit 1; }
  D=gpurun_out/kt_${TAG}_${WL}
  KS=10; [ $WL = hf10m ] && KS=4
  # --family-times: the process also measures every launch with HIP event pairs in three (five) more frames of the same kind and
  # prints the per-family averages (bench_under_rocprof.json: hip_event_family_times) -- the figure bench.py's roofline block
  # is built on, from the SAME process the CSV describes
  # --no-roofline: only frames that run like the timed ones (warm-up + timed steps, queued back to back on the two slots; the
  # traversal mode is pinned above, so no calibration batches either) reach the CSV: its average launch durations are the ones
  # bench.py's live avg_launch_ms measures with HIP events in its overlapped timing frames
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 bench.py --workload $WL --steps $KS --warmup 1 --no-cpu-baseline --no-roofline --family-times > gpurun_out/${TAG}_${WL}_kt.log 2>&1 || { echo "kernel trace $WL failed"; tail -5 gpurun_out/${TAG}_${WL}_kt.log; exit 1; }
  cp "$(find $D -name '*kernel_stats.csv' | head -1)" profiles/${TAG}_${WL}_kernel_stats.csv || exit 1
  # last_frames: the trace's own durations of the launches of the 3 (5) event-timed frames alone
  python3 tools/family_stats.py profiles/${TAG}_${WL}_kernel_stats.csv $((KS + 1 + 3 + 2 * (KS > 4))) "$(find $D -name '*kernel_trace.csv' | head -1)" $((3 + 2 * (KS > 4))) > profiles/${TAG}_${WL}_family_stats.json || exit 1
  grep "^{\"metric\"" gpurun_out/${TAG}_${WL}_kt.log | tail -n 1 > profiles/${TAG}_${WL}_bench_under_rocprof.json
  rm -rf $D gpurun_out/pmc_${TAG}_${WL}
done
unset RTHIP_REFILL_MODE
timeout -k 10 300 python3 bench.py > profiles/${TAG}_hf10m_bench.json 2> gpurun_out/${TAG}_hf10m_bench.err || { echo "bench hf10m failed"; tail -5 gpurun_out/${TAG}_hf10m_bench.err; exit 1; }
timeout -k 10 300 python3 bench.py --workload soup1m > profiles/${TAG}_soup1m_bench.json 2> gpurun_out/${TAG}_soup1m_bench.err || { echo "bench soup failed"; exit 1; }
timeout -k 10 300 python3 bench.py --workload cube > profiles/${TAG}_cube_bench.json 2> gpurun_out/${TAG}_cube_bench.err || { echo "bench cube failed"; exit 1; }
timeout -k 10 300 python3 bench.py --workload bunny > profiles/${TAG}_bunny_bench.json 2> gpurun_out/${TAG}_bunny_bench.err || { echo "bench bunny failed"; exit 1; }
timeout -k 10 400 python3 bench.py --workload hf8k --steps 2 --warmup 1 --cpu-seconds 6 > profiles/${TAG}_hf8k_bench.json 2> gpurun_out/${TAG}_hf8k_bench.err || { echo "bench hf8k failed"; exit 1; }
mkdir -p gpurun_out/profiles_${TAG} && cp profiles/${TAG}_* profiles/traffic.json gpurun_out/profiles_${TAG}/
cat profiles/${TAG}_hf10m_bench.json
