timeout -k 10 200 python3 -u tests/tune_probe.py soup1 soup8 bunny
