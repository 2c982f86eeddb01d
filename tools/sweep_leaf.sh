# frame time against the builder's SAH traversal cost and leaf size (scene options traversalCost, maxLeafTriangles)
for WL in soup1m hf10m4; do
for v in "1,4" "1.5,4" "2,4" "3,4" "2,8" "3,8" "5,8"; do
  ct=${v%,*}; ml=${v#*,}
  echo "== $WL traversalCost $ct maxLeafTriangles $ml"
  RT_PROBE_OPTS="traversalCost=$ct,maxLeafTriangles=$ml" RT_PROBE_ALONE= timeout -k 10 200 python tools/kernel_probe.py $WL 6 2>&1 | grep -E "pipelining=1|nodes/ray"
done; done
