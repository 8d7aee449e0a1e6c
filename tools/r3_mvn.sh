timeout 900 python3 -m pytest tests/test_gpu_mvn.py -x -q 2>&1 | tail -5
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_specialised.py -x -q -k "gp_" 2>&1 | tail -25
