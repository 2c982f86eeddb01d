# frame time of a workload against the refill / leaf-step thresholds of the traversal kernel
# (RTHIP_TUNE = refill,leafLo,blocksPerCU,steal,overlapShadow,pipeline,leafHi)
WL=${1:-hf10m}; FR=${2:-2}
for cfg in "4 24" "4 16" "4 32" "8 24" "2 24" "4 4" "24 24"; do
  set -- $cfg
  export RTHIP_TUNE=16,$1,6,0,1,2,$2
  echo "== leaf lo,hi = $1,$2"
  timeout -k 10 120 python tools/kernel_probe.py $WL $FR 2>&1 | grep "pipelining=1"
done
