#!/bin/bash
# A/B of kernel-library builds on ONE box (box-to-box spread is +-2 %, more than most code-shape effects): the variants
# raytracer_amd/librthip_<name>.so are copied over librthip.so in turn, round after round, and each runs the probe workloads.
#   usage (through gpurun): bash tools/ab_libs.sh "<name> <name> ..." [rounds] ["<workload> <frames> <mode>;..."]
# Build a variant with the Makefile's flags plus whatever is being tried: hipcc $(HIPFLAGS) ... -o librthip_<name>.so csrc/rthip.hip
set -u
VARIANTS=${1:?variants}
ROUNDS=${2:-2}
WORK=${3:-"hf10m 4 2;soup1m 30 0"}
mkdir -p gpurun_out/ab
cp raytracer_amd/librthip.so gpurun_out/ab/keep.so
for r in $(seq 1 $ROUNDS); do
  for v in $VARIANTS; do
    cp raytracer_amd/librthip_$v.so raytracer_amd/librthip.so || exit 1
    IFS=';' read -ra WL <<< "$WORK"
    for w in "${WL[@]}"; do
      set -- $w
      RTHIP_LIST_BLOCKS=${LIST_BLOCKS:-3} RTHIP_REFILL_MODE=$3 timeout -k 10 300 python tools/kernel_probe.py $1 $2 > gpurun_out/ab/o.txt 2>&1 || { tail -3 gpurun_out/ab/o.txt; cp gpurun_out/ab/keep.so raytracer_amd/librthip.so; exit 1; }
      echo "$v $(grep -E 'pipelining=1' gpurun_out/ab/o.txt | cut -c1-62)"
    done
  done
done
cp gpurun_out/ab/keep.so raytracer_amd/librthip.so
