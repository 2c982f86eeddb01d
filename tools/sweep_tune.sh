timeout -k 10 100 python3 bench.py --no-cpu-baseline | cut -c1-160
timeout -k 10 100 python3 bench.py --no-cpu-baseline --workload bunny | cut -c1-160
RTHIP_TUNE=16,4,6,0,1,0 timeout -k 10 100 python3 bench.py --no-cpu-baseline | cut -c1-160
