#!/bin/bash
# One profiling round on the GPU box: bench lines, rocprofv3 kernel-trace stats and PMC passes for the headline workload
# (cube) and the incoherent one (soup1m); summaries land in profiles/ under the given tag.
#   usage: tools/profile_round.sh <tag>        (run through gpurun; every GPU step has its own time limit)
set -u
TAG=${1:-r01}
export TMPDIR=/tmp
mkdir -p gpurun_out profiles
for WL in cube soup1m; do
  bash tools/profile_pmc.sh $TAG $WL 3 > gpurun_out/${TAG}_${WL}_pmc.log 2>&1 || { echo "pmc $WL failed"; tail -5 gpurun_out/${TAG}_${WL}_pmc.log; exit 1; }
  D=gpurun_out/kt_${TAG}_${WL}
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 bench.py --workload $WL --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_${WL}_kt.log 2>&1 || { echo "kernel trace $WL failed"; tail -5 gpurun_out/${TAG}_${WL}_kt.log; exit 1; }
  cp "$(find $D -name '*kernel_stats.csv' | head -1)" profiles/${TAG}_${WL}_kernel_stats.csv || exit 1
  grep "^{\"metric\"" gpurun_out/${TAG}_${WL}_kt.log | tail -n 1 > profiles/${TAG}_${WL}_bench_under_rocprof.json
done
timeout -k 10 300 python3 bench.py > profiles/${TAG}_cube_bench.json 2> gpurun_out/${TAG}_cube_bench.err || { echo "bench cube failed"; tail -5 gpurun_out/${TAG}_cube_bench.err; exit 1; }
timeout -k 10 300 python3 bench.py --workload soup1m > profiles/${TAG}_soup1m_bench.json 2> gpurun_out/${TAG}_soup1m_bench.err || { echo "bench soup failed"; exit 1; }
timeout -k 10 300 python3 bench.py --workload bunny > profiles/${TAG}_bunny_bench.json 2> gpurun_out/${TAG}_bunny_bench.err || { echo "bench bunny failed"; exit 1; }
timeout -k 10 400 python3 bench.py --workload hf10m --steps 3 --warmup 1 > profiles/${TAG}_hf10m_bench.json 2> gpurun_out/${TAG}_hf10m_bench.err || { echo "bench hf10m failed"; exit 1; }
cp profiles/*.json profiles/*.csv gpurun_out/ 2>/dev/null
mkdir -p gpurun_out/profiles_${TAG} && cp profiles/${TAG}_* profiles/traffic.json gpurun_out/profiles_${TAG}/
cat profiles/${TAG}_cube_bench.json
