#!/bin/bash
# Runs tests/tune_probe.py once per RTHIP_TUNE setting given on the command line ("refill,leaf[,blocksPerCU[,steal[,overlap[,pieces]]]]"),
# e.g.  gpurun -- 'bash tools/sweep_tune.sh 16,4 8,4 16,4,4 > gpurun_out/sweep.log'.  Workloads: env PROBE (default "soup1 soup8 bunny").
# Experiments only; every GPU step has its own time limit and the sweep stops at the first failure.
for t in "$@"; do
  RTHIP_TUNE=$t timeout -k 10 240 python3 -u tests/tune_probe.py ${PROBE:-soup1 soup8 bunny} || exit 1
done
