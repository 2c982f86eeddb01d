cp raytracer_amd/ab/librthip_base.so raytracer_amd/librthip.so; echo BASE; timeout -k 10 200 python3 -u tests/tune_probe.py soup1 soup8 bunny || exit 1
cp raytracer_amd/ab/librthip_w7.so raytracer_amd/librthip.so; echo "W7 blocks 7"; RTHIP_TUNE=16,4,7 timeout -k 10 200 python3 -u tests/tune_probe.py soup1 soup8 bunny || exit 1
echo "W7 blocks 6"; RTHIP_TUNE=16,4,6 timeout -k 10 200 python3 -u tests/tune_probe.py soup1 soup8 bunny
