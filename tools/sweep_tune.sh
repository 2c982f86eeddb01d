timeout -k 10 200 python3 -u tests/tune_probe.py soup1 soup8 bunny && RTHIP_TUNE=16,4,6,0,0 timeout -k 10 200 python3 -u tests/tune_probe.py soup1 soup8 bunny
