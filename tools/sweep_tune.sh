for leaf in 1 2 3 4 6 8; do echo "maxLeafTriangles $leaf"; RT_maxLeafTriangles=$leaf timeout -k 10 200 python3 -u tests/tune_probe.py soup8 bunny || exit 1; done
