#!/bin/bash
# Collects rocprofv3 PMC counters for bench.py, one pass per run (separate --pmc passes, no trace domains mixed in),
# then summarises per kernel into gpurun_out/pmc_<tag>_<workload>.json and merges the per-launch HBM traffic into
# profiles/traffic.json (read by bench.py for roofline.traffic).   usage: tools/profile_pmc.sh <tag> <workload> [steps]
set -u
TAG=${1:-r01}; WL=${2:-cube}; STEPS=${3:-3}
OUT=gpurun_out/pmc_${TAG}_${WL}
mkdir -p $OUT
export TMPDIR=/tmp
run() { # name counters...
  local name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py --workload $WL --steps $STEPS --warmup 1 --no-cpu-baseline --no-roofline > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT/$name.log; return 1; }
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_SALU &&
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM GRBM_GUI_ACTIVE &&
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum &&
run tcc2 FETCH_SIZE &&
run tcc3 WRITE_SIZE &&
python3 tools/summarize_pmc.py $OUT --traffic $WL profiles/traffic.json > profiles/${TAG}_${WL}_pmc.json && python3 -c "
import json;d=json.load(open('profiles/${TAG}_${WL}_pmc.json'));print(json.dumps(d['_per_kernel_family']))"
