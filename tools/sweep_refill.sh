# frame time against the refill rule of the traversal launches: RTHIP_TUNE fields 8-11 = low threshold of shadow launches, of dynamic
# primary launches, the high threshold, the patience (node steps without a newly idle lane)
#   usage: tools/sweep_refill.sh <workload> <frames> "<rs,rp,hi,patience> ..."
WL=${1:-hf10m}; FR=${2:-3}; SETS=${3:-"16,32,64,0 16,32,56,1 16,32,56,2 16,32,56,4 16,32,56,8 16,16,56,2"}
for v in $SETS; do
  export RTHIP_FORCE_GROUP=0 RTHIP_TUNE=16,4,6,0,1,2,24,$v
  echo "== $WL shadow lo, primary lo, hi, patience: $v"
  RT_PROBE_ALONE= timeout -k 10 300 python tools/kernel_probe.py $WL $FR 2>&1 | grep -E "pipelining=1" || exit 1
done
