#!/bin/bash
# A/B of environment settings on ONE box, interleaved round after round (see tools/ab_libs.sh for why).
#   usage (through gpurun): bash tools/ab_env.sh "<VAR=val[,VAR=val]> <...> ..." [rounds] ["<workload> <frames> <mode>;..."]     ("-" = no setting)
set -u
CONFIGS=${1:?configs}
ROUNDS=${2:-2}
WORK=${3:-"hf10m 4 2;soup1m 30 0"}
mkdir -p gpurun_out/ab
for r in $(seq 1 $ROUNDS); do
  for c in $CONFIGS; do
    IFS=';' read -ra WL <<< "$WORK"
    for w in "${WL[@]}"; do
      set -- $w
      E=$(echo "$c" | tr '+' ' '); [ "$c" = "-" ] && E="RT_NONE=1"
      env RTHIP_LIST_BLOCKS=${LIST_BLOCKS:-3} RTHIP_REFILL_MODE=$3 $E timeout -k 10 300 python tools/kernel_probe.py $1 $2 > gpurun_out/ab/o.txt 2>&1 || { tail -3 gpurun_out/ab/o.txt; exit 1; }
      echo "$c $(grep -E 'pipelining=1' gpurun_out/ab/o.txt | cut -c1-62)"
    done
  done
done
