export RT_PROBE_ALONE=
for cfg in "27 -1" "27 2" "27 4" "26 -1" "28 -1" "25 -1"; do
  set -- $cfg
  export RTHIP_PATHCAP_LOG2=$1
  if [ "$2" = "-1" ]; then unset RTHIP_GROUP_LG; else export RTHIP_GROUP_LG=$2; fi
  echo "== pathcap 2^$1 group_lg $2"
  timeout -k 10 120 python tools/kernel_probe.py hf10m 2 2>&1 | grep "pipelining=1"
done
