export RTHIP_REFILL_MODE=1
for v in "4,24" "4,32" "4,40" "8,32" "8,48" "16,48"; do lo=${v%,*}; hi=${v#*,}
  export RTHIP_TUNE=16,$lo,6,0,1,2,$hi
  echo "== hf10m leaf lo $lo hi $hi"
  RT_PROBE_ALONE= timeout -k 10 300 python tools/kernel_probe.py hf10m 3 2>&1 | grep -E "pipelining=1" || exit 1
done
for h in 48 52 56 60; do
  export RTHIP_TUNE=16,4,6,0,1,2,24,16,32,$h
  echo "== hf10m refill hi $h"
  RT_PROBE_ALONE= timeout -k 10 300 python tools/kernel_probe.py hf10m 3 2>&1 | grep -E "pipelining=1" || exit 1
done
