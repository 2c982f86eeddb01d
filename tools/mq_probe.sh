# Probe: more batches in flight (several contexts) against the hardware-queue limit (GPU_MAX_HW_QUEUES; HIP's default is 4).
set -e
W=${1:-soup1m}; F=${2:-40}
for q in 4 8; do
  for R in 1 2 3; do
    echo "queues=$q"; GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python -u tools/multi_ctx_probe.py $W $R $F
  done
done
