timeout 900 python3 -m pytest tests/test_gpu_collective.py tests/test_gpu_two_ranks.py -x -q 2>&1 | tail -15
