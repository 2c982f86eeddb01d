for spp in 4 16 64; do for v in "16,32,64,0" "56,56,56,0" "40,40,56,0"; do
  export RT_PROBE_SPP=$spp RTHIP_TUNE=16,4,6,0,1,2,24,$v
  echo "== soup1m spp $spp refill $v"
  RT_PROBE_ALONE= timeout -k 10 300 python tools/kernel_probe.py soup1m 3 2>&1 | grep -E "pipelining=1" || exit 1
done; done
