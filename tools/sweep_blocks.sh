# frame time of a workload against the number of persistent trace blocks per CU (RTHIP_TUNE = refill,leaf,blocksPerCU)
WL=${1:-hf10m}; FR=${2:-2}
for b in 6 4 3 2 1; do
  export RTHIP_TUNE=16,4,$b
  echo "== blocks per CU $b"
  timeout -k 10 120 python tools/kernel_probe.py $WL $FR 2>&1 | grep "pipelining\|kernels alone\|lane use"
done
