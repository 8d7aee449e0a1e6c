timeout 900 python3 -m pytest tests/test_gpu_mvn.py -x -q 2>&1 | tail -25
