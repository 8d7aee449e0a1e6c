timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "exact_data_path_at_config4" 2>&1 | tail -25
